#!/usr/bin/env python
"""The thin level-2 products alone: [32 x 1140 x 1140] x [32 x 1140 x N], N = 20 / 40 / 114, NN and TN, automatic tile choice and
forced configurations (cgc_gemm_tuning: 1..6 = 128x128, 128x64, 64x128, 64x64, 128x32, 32x128; +10 pipelined kernel).
env CGC_GEMM_SPLIT=0 switches the tail split off (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
B, M, Kd = 32, 1140, 1140
A = torch.randn(B, M, Kd, device=dev)
warm = torch.randn(8192, 8192, device=dev)


COLD = os.environ.get('THIN_COLD', '0') == '1'      # THIN_COLD=1: 1 GB written between the calls (nothing of A left in L2 / Infinity Cache)
flush = torch.empty(256 * 1024 * 1024, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    (warm @ warm)                     # keeps the clock up between the short launches
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if COLD:
            flush.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / reps * 1e3


for N in (20, 40, 114):
    Bm = torch.randn(B, Kd, N, device=dev)
    C = torch.empty(B, M, N, device=dev)
    for tA in (False, True):
        call = lambda: K.gemm(A, Bm, C, M, N, Kd, tA, False, Kd, N, N, 1.0, 0.0, None, B, M * Kd, Kd * N, M * N)
        K.lib.cgc_gemm_tuning(0)
        line = ['N %3d %s: auto %6.1f us' % (N, 'TN' if tA else 'NN', timeit(call))]
        for cfg in (11, 12, 14, 15):
            K.lib.cgc_gemm_tuning(cfg)
            line.append('%s %6.1f' % ({11: '128x128', 12: '128x64', 14: '64x64', 15: '128x32'}[cfg], timeit(call)))
        K.lib.cgc_gemm_tuning(0)
        print('  '.join(line), ' [cold=%d lib=%s]' % (COLD, os.path.basename(kernels.lib_path())))
print('read of A alone (sum): %.1f us' % timeit(lambda: A.sum()))
