#!/usr/bin/env python
"""Does ONE workgroup per CU keep its matrix pipes busy?  128x128 tiles, exactly 256 / 512 / 1024 tiles, K = 4096: with perfect
intra-workgroup pipelining the 256-tile product runs at the MFMA rate of one wave per SIMD (the pipe does not care how many
waves feed it); whatever is missing is latency the k loop of a lone workgroup does not hide."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

K = kernels.get()
dev = 'cuda:0'


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


K.lib.cgc_gemm_tuning(11)            # 128x128, pipelined
for tA, tB in ((False, False), (False, True), (True, False)):
    for M, N in ((2048, 2048), (4096, 2048), (4096, 4096), (8192, 4096), (8192, 8192)):
        Kd = 4096
        A = torch.randn((Kd, M) if tA else (M, Kd), device=dev)
        B = torch.randn((N, Kd) if tB else (Kd, N), device=dev)
        C = torch.empty(M, N, device=dev)
        t = timeit(lambda: K.gemm(A, B, C, M, N, Kd, tA, tB, A.shape[1], B.shape[1], N))
        tiles = (M // 128) * (N // 128)
        print('%s%s %5d x %5d x %d: %4d tiles (%.1f per CU)  %8.1f us  %6.1f TF/s  %.2f us per k-step per workgroup-round' % (
            'T' if tA else 'N', 'T' if tB else 'N', M, N, Kd, tiles, tiles / 256, t, 2.0 * M * N * Kd / t / 1e6,
            t / (Kd / 32) / max(1, -(-tiles // 512))))
K.lib.cgc_gemm_tuning(0)
