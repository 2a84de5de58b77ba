#!/usr/bin/env python
"""The dominant GEMM alone, on the step's flat shapes (58761 x 1140 x 1140, rows on 1152-float strides: the FAST 128 x 128 kernel),
NN / NT / TN, each timed over a few launches after a long warm-up (the clock ramps over milliseconds).  CGC_LIB selects a variant
library (tools/variant_lib.sh).  usage: python tools/gemm_quick.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, N, Kd, LD = 58761, 1140, 1140, 1152
torch.manual_seed(0)
big = torch.randn(M, LD, device=dev)
big2 = torch.randn(M, LD, device=dev)
sq = torch.randn(LD, LD, device=dev)
out_big = torch.empty(M, LD, device=dev)
out_sq = torch.empty(LD, LD, device=dev)
ws = torch.empty(int(K.lib.cgc_gemm_ws_floats()), device=dev)
cases = [
    ('NN [M,K]x[K,N]', lambda: K.gemm(big, sq, out_big, M, N, Kd, False, False, LD, LD, LD), 2.0 * M * N * Kd),
    ('NT [M,K]x[N,K]^T', lambda: K.gemm(big, sq, out_big, M, N, Kd, False, True, LD, LD, LD), 2.0 * M * N * Kd),
]
for name, fn, fl in cases:
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print('%-20s %8.1f us  %6.1f TF  (%.3f of 157.3)   lib=%s' % (name, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3, os.path.basename(kernels.lib_path())))
