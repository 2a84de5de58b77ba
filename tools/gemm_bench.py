#!/usr/bin/env python
"""Micro-benchmark of cgc_gemm_f32 on the shapes of the C3 workload (and square references), next to torch.matmul
(rocBLAS/hipBLASLt) as a same-hardware reference point.  GPU only.  usage: python tools/gemm_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def flat(M, N, Kd, tA, tB, name):
    A = torch.randn((Kd, M) if tA else (M, Kd), device=dev)
    B = torch.randn((N, Kd) if tB else (Kd, N), device=dev)
    C = torch.empty(M, N, device=dev)
    ms = timeit(lambda: K.gemm(A, B, C, M, N, Kd, tA, tB, A.shape[1], B.shape[1], N))
    a, b = (A.t() if tA else A), (B.t() if tB else B)
    ms_t = timeit(lambda: torch.matmul(a, b, out=C))
    fl = 2.0 * M * N * Kd
    print('%-34s M=%6d N=%5d K=%6d %s%s  ours %8.1f us %6.1f TF | torch %8.1f us %6.1f TF' % (
        name, M, N, Kd, 'T' if tA else 'N', 'T' if tB else 'N', ms * 1e3, fl / ms / 1e9, ms_t * 1e3, fl / ms_t / 1e9))


def ragged(C, D, counts, mode, name):
    n = int(sum(counts))
    gptr = torch.tensor(np.cumsum([0] + list(counts)), dtype=torch.int32, device=dev)
    Bn, nmax = len(counts), int(max(counts))
    S = torch.randn(n, C, device=dev)
    if mode == 'K':      # out[b] = S_b^T X_b
        X = torch.randn(n, D, device=dev)
        out = torch.empty(Bn, C, D, device=dev)
        ms = timeit(lambda: K.gemm(S, X, out, C, D, 0, True, False, C, D, D, 1.0, 0.0, None, Bn, 0, 0, C * D, gptr, 2, nmax, n))
    elif mode == 'M_NN':  # Y_b = S_b G_b
        G = torch.randn(Bn, C, D, device=dev)
        out = torch.empty(n, D, device=dev)
        ms = timeit(lambda: K.gemm(S, G, out, 0, D, C, False, False, C, D, D, 1.0, 0.0, None, Bn, 0, C * D, 0, gptr, 1, nmax, n))
    else:                 # Y_b = S_b G_b^T
        G = torch.randn(Bn, D, C, device=dev)
        out = torch.empty(n, D, device=dev)
        ms = timeit(lambda: K.gemm(S, G, out, 0, D, C, False, True, C, C, D, 1.0, 0.0, None, Bn, 0, C * D, 0, gptr, 1, nmax, n))
    fl = 2.0 * n * C * D
    print('%-34s n=%6d C=%5d D=%5d ragged-%s  ours %8.1f us %6.1f TF' % (name, n, C, D, mode, ms * 1e3, fl / ms / 1e9))


if __name__ == '__main__':
    rng = np.random.RandomState(0)
    counts = rng.randint(1440, 2161, size=32)
    n = int(counts.sum())
    for sz in (2048, 4096):
        for tA, tB in ((False, False), (False, True), (True, False)):
            flat(sz, sz, sz, tA, tB, 'square')
    flat(n, 1140, 1180, False, True, 'lin fwd  cat @ W^T')
    flat(n, 1180, 1140, False, False, 'lin bwd  dAssign @ W')
    flat(1140, 1180, n, True, False, 'lin dW   dAssign^T @ cat (no split)')
    flat(n, 1140, 20, False, False, 'gcn3     agg @ W3')
    flat(n, 20, 1140, False, True, 'gcn3 bwd dh @ W3^T')
    flat(n, 20, 16, False, False, 'gcn1     agg @ W1')
    ragged(1140, 1140, counts, 'K', 'A2 = S^T (A S)')
    ragged(1140, 60, counts, 'K', 'X2 = S^T X')
    ragged(1140, 1140, counts, 'M_NN', 'dP = S dA2')
    ragged(1140, 1140, counts, 'M_NT', 'dS = P dA2^T')
    ragged(1140, 60, counts, 'M_NN', 'dX = S dX2')
    # level 2, strided batch of 32
    import itertools
    for (Mm, Nn, Kk, tA, tB) in [(1140, 1140, 114, False, True), (1140, 1140, 140, False, True), (1140, 114, 1140, False, False), (1140, 114, 1140, True, False), (114, 114, 1140, True, False), (1140, 20, 1140, True, False), (1140, 60, 1140, True, False)]:
        Am = torch.randn(32, *((Kk, Mm) if tA else (Mm, Kk)), device=dev)
        Bm = torch.randn(32, *((Nn, Kk) if tB else (Kk, Nn)), device=dev)
        Cm = torch.empty(32, Mm, Nn, device=dev)
        ms = timeit(lambda: K.gemm(Am, Bm, Cm, Mm, Nn, Kk, tA, tB, Am.shape[2], Bm.shape[2], Nn, 1.0, 0.0, None, 32, Am.shape[1] * Am.shape[2], Bm.shape[1] * Bm.shape[2], Mm * Nn))
        a_, b_ = (Am.transpose(1, 2) if tA else Am), (Bm.transpose(1, 2) if tB else Bm)
        ms_t = timeit(lambda: torch.bmm(a_, b_, out=Cm))
        fl = 2.0 * 32 * Mm * Nn * Kk
        print('level-2 %s%s M=%4d N=%4d K=%4d b=32          ours %8.1f us %6.1f TF | torch.bmm %8.1f us %6.1f TF' % ('T' if tA else 'N', 'T' if tB else 'N', Mm, Nn, Kk, ms * 1e3, fl / ms / 1e9, ms_t * 1e3, fl / ms_t / 1e9))
    A = torch.randn(32, 1140, 1140, device=dev)
    for F in (60, 20, 114):
        X = torch.randn(32, 1140, F, device=dev)
        C = torch.empty(32, 1140, F, device=dev)
        ms = timeit(lambda: K.gemm(A, X, C, 1140, F, 1140, False, False, 1140, F, F, 1.0, 0.0, None, 32, 1140 * 1140, 1140 * F, 1140 * F))
        ms_t = timeit(lambda: torch.bmm(A, X, out=C))
        fl = 2.0 * 32 * 1140 * 1140 * F
        print('level-2 A~ @ x  F=%3d                 ours %8.1f us %6.1f TF | torch.bmm %8.1f us %6.1f TF' % (
            F, ms * 1e3, fl / ms / 1e9, ms_t * 1e3, fl / ms_t / 1e9))
