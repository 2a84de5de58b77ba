#!/usr/bin/env python
"""Micro-benchmark of the DenseJK kernels (bi-LSTM over 3 layer embeddings + attention) at the three level sizes of C3.
usage: python tools/jk_bench.py [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
C, H = 20, 30
rows = [int(a) for a in sys.argv[1:]] or [57711, 36480, 3648]
torch.manual_seed(0)
lstm = []
for d in range(2):
    lstm += [torch.randn(4 * H, C, device=dev) * 0.2, torch.randn(4 * H, H, device=dev) * 0.2,
             torch.randn(4 * H, device=dev) * 0.1, torch.randn(4 * H, device=dev) * 0.1]
w_att, b_att = torch.randn(1, 2 * H, device=dev) * 0.2, torch.zeros(1, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n in rows:
    npad = -(-n // 1024) * 1024
    xs, dout = torch.randn(n, 3 * C, device=dev), torch.randn(n, C, device=dev)
    out = torch.empty(n, C, device=dev)
    HS, CS = torch.empty(6 * H, npad, device=dev), torch.empty(6 * H, npad, device=dev)
    ng, ni, ktot = 4 * H + 1, C + 2 * H + 1, 3 * npad
    dxs = torch.empty_like(xs)
    DGT, INT = torch.empty(2, ng, ktot, device=dev), torch.empty(2, ni, ktot, device=dev)
    DHC = torch.empty(2, 2, H, npad, device=dev)
    tf = timeit(lambda: K.jk_fwd(xs, n, npad, C, lstm, w_att, b_att, out, HS, CS))
    tb = timeit(lambda: K.jk_bwd(xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, DGT, INT, DHC))
    G = torch.empty(2, ng, ni, device=dev)
    tp = timeit(lambda: K.jk_bwd_params(xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, G))
    fl = 2.0 * 4 * H * (C + H) * 3 * 2 * n          # gate products, forward
    print('rows %6d: fwd %7.1f us (%5.1f TFLOP/s)   bwd (staged gradients, without its GEMMs) %7.1f us   bwd with parameter '
          'gradients in-kernel %7.1f us' % (n, tf, fl / tf / 1e6, tb, tp))
