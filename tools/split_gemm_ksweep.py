#!/usr/bin/env python
"""Flat NN product [58761, K] x [K, 1140] in both GEMM modes for several K: the slope is the cost of a k-tile sweep over all output
tiles, the intercept what a launch pays besides (per-tile prologue + epilogue x rounds, tail round, launch).  usage: [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, N = 58761, 1140
torch.manual_seed(0)
out = torch.empty(M, 1152, device=dev)
rows = []
for Kd in (288, 576, 1152, 2304):
    X = torch.randn(M, Kd, device=dev)
    W = torch.randn(Kd, 1152, device=dev) * 0.05
    line = 'K = %4d' % Kd
    for mode in (0, 1):
        K.gemm_mode = mode
        fn = lambda: K.gemm(X, W, out, M, N, Kd, False, False, Kd, 1152, 1152)
        for _ in range(25):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        line += '   %s %8.1f us (%6.1f TF)' % ('exact' if mode == 0 else 'split', ms * 1e3, 2.0 * M * N * Kd / ms / 1e9)
        rows.append((Kd, mode, ms * 1e3))
    K.gemm_mode = 0
    print(line)
for mode in (0, 1):
    pts = [(k, t) for k, m, t in rows if m == mode]
    slope = (pts[-1][1] - pts[1][1]) / (pts[-1][0] - pts[1][0])
    print('%s: %.3f us per unit of K (K = 576 -> 2304), intercept %.1f us' % ('exact' if mode == 0 else 'split', slope, pts[1][1] - slope * pts[1][0]))
