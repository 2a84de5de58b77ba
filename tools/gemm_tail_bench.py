#!/usr/bin/env python
"""The six dominant contractions of a step (C1 = 1140) at 4 / 8 / 16 / 32 graphs per GPU, with and without the tail split of the
128 x 128 kernel (cgc_gemm_f32_ws): time per launch and TFLOP/s.  usage: tools/gemm_tail_bench.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels, ops  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.graph import BatchGraph  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
C, LD = 1140, 1152


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def wide(n):
    return torch.randn(n, LD, device=dev)[:, :C]


for B in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32]:
    ds_ = SyntheticCellGraphs(B, 1800, 16, base_seed=0)
    g = BatchGraph.from_batch(Batch.from_data_list([ds_[i] for i in range(B)]).to(dev))
    n = g.n
    x, s, p, out = wide(n), wide(n), wide(n), wide(n)
    w = wide(C)
    x12, w12 = torch.randn(n, 40, device=dev), torch.randn(40, LD, device=dev)[:, :C]
    dao, dxo, emb = torch.randn(B, C, C, device=dev), torch.randn(B, C, 60, device=dev), torch.randn(n, 60, device=dev)
    ao, dw = torch.empty(B, C, C, device=dev), torch.empty(C, C, device=dev)
    fl = 2.0 * n * C * C
    cases = [
        ('Linear fwd  NN flat + extra 40', lambda: K.gemm(x, w, out, n, C, C, False, False, LD, LD, LD, 1.0, 0.0, None,
                                                          extra=[(x12, w12, 40, LD, 40, 0, 0)]), fl + 2.0 * n * 40 * C),
        ('Linear dx   NN flat', lambda: K.gemm(x, w, out, n, C, C, False, False, LD, LD, LD), fl),
        ('Linear dW   TN row-split', lambda: ops.gemm_tn_rows(x, LD, C, s, LD, C, n, dw), fl),
        ('S^T P       TN ragged K', lambda: K.gemm(s, p, ao, C, C, 0, True, False, LD, LD, C, 1.0, 0.0, None, B, 0, 0, C * C,
                                                   g.gptr, 2, g.nmax, n), fl),
        ('dP = S dA   NN ragged M', lambda: K.gemm(s, dao, out, 0, C, C, False, False, LD, C, LD, 1.0, 0.0, None, B, 0, C * C, 0,
                                                   g.gptr, 1, g.nmax, n), fl),
        ('dS += P dA^T + X dX^T  NT ragged M', lambda: K.gemm(p, dao, out, 0, C, C, False, True, LD, C, LD, 1.0, 1.0, None, B, 0,
                                                              C * C, 0, g.gptr, 1, g.nmax, n,
                                                              extra=[(emb, dxo, 60, 60, 60, 0, C * 60)]), fl + 2.0 * n * 60 * C),
    ]
    print('B = %d  (%d rows)' % (B, n))
    tot = [0.0, 0.0]
    for name, fn, f in cases:
        r = []
        for i, split in enumerate((False, True)):
            K.tail_split = split
            ms = timeit(fn)
            tot[i] += ms
            r.append('%7.1f us %6.1f TF' % (ms * 1e3, f / ms / 1e9))
        print('  %-38s whole %s | split %s' % (name, r[0], r[1]))
    print('  %-38s whole %7.1f us           | split %7.1f us' % ('sum', tot[0] * 1e3, tot[1] * 1e3))
K.tail_split = True
