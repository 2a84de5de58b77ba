#!/usr/bin/env python
"""time(K) at fixed 4096 x 4096 outputs (1024 tiles = two full rounds of 512 resident workgroups): the intercept of the linear
fit is the per-launch fixed cost (dispatch ramp + prologues + epilogues + drain), the slope the steady-state MFMA rate."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (M, N) in ((4096, 4096), (8192, 4096), (58000, 1152)):
    for tA, tB in ((False, True), (False, False), (True, False)):
        ks, ts = [512, 1024, 2048, 4096], []
        for Kd in ks:
            A = torch.randn((Kd, M) if tA else (M, Kd), device=dev)
            B = torch.randn((N, Kd) if tB else (Kd, N), device=dev)
            C = torch.empty(M, N, device=dev)
            ts.append(timeit(lambda: K.gemm(A, B, C, M, N, Kd, tA, tB, A.shape[1], B.shape[1], N)))
        slope, icpt = np.polyfit(ks, ts, 1)
        tf = 2.0 * M * N / slope / 1e6
        print('M=%5d N=%5d %s%s  us at K=512..4096: %s | fit: %.1f us fixed + %.4f us/k  => steady %.1f TF (%.3f of peak)' % (
            M, N, 'T' if tA else 'N', 'T' if tB else 'N', ' '.join('%.0f' % t for t in ts), icpt, slope, tf, tf / 157.3))
