#!/usr/bin/env python
"""A handful of cgc_gemm_f32 launches for counter passes: square NN / NT / TN 4096 and the C3 workload's flat and ragged shapes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = sys.argv[2] if len(sys.argv) > 2 else 'all'


def flat(M, N, Kd, tA, tB):
    A = torch.randn((Kd, M) if tA else (M, Kd), device=dev)
    B = torch.randn((N, Kd) if tB else (Kd, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(reps):
        K.gemm(A, B, C, M, N, Kd, tA, tB, A.shape[1], B.shape[1], N)


if which in ('all', 'square'):
    flat(4096, 4096, 4096, False, False)
    flat(4096, 4096, 4096, False, True)
    flat(4096, 4096, 4096, True, False)
if which in ('all', 'work'):
    rng = np.random.RandomState(0)
    counts = rng.randint(1440, 2161, size=32)
    n = int(counts.sum())
    gptr = torch.tensor(np.cumsum([0] + list(counts)), dtype=torch.int32, device=dev)
    nmax = int(counts.max())
    S, P = torch.randn(n, 1152, device=dev)[:, :1140], torch.randn(n, 1152, device=dev)[:, :1140]
    G = torch.randn(32, 1140, 1140, device=dev)
    out = torch.empty(32, 1140, 1140, device=dev)
    Y = torch.empty(n, 1152, device=dev)[:, :1140]
    for _ in range(reps):
        K.gemm(S, P, out, 1140, 1140, 0, True, False, 1152, 1152, 1140, 1.0, 0.0, None, 32, 0, 0, 1140 * 1140, gptr, 2, nmax, n)
        K.gemm(S, G, Y, 0, 1140, 1140, False, False, 1152, 1140, 1152, 1.0, 0.0, None, 32, 0, 1140 * 1140, 0, gptr, 1, nmax, n)
        K.gemm(P, G, Y, 0, 1140, 1140, False, True, 1152, 1140, 1152, 1.0, 1.0, None, 32, 0, 1140 * 1140, 0, gptr, 1, nmax, n)
torch.cuda.synchronize()
