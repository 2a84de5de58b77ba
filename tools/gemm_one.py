#!/usr/bin/env python
"""Run one cgc_gemm_f32 shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K tA tB [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

M, N, Kd, tA, tB = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = 'cuda:0'
K = kernels.get()
A = torch.randn((Kd, M) if tA else (M, Kd), device=dev)
B = torch.randn((N, Kd) if tB else (Kd, N), device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(reps):
    K.gemm(A, B, C, M, N, Kd, bool(tA), bool(tB), A.shape[1], B.shape[1], N)
torch.cuda.synchronize()
