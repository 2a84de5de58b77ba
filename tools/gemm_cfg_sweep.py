#!/usr/bin/env python
"""Tile-shape sweep over the products of one training step (gpurun_out/gemm_shapes.json from tools/gemm_log_shapes.py): for
every distinct non-dominant shape, the time of the automatic choice and of each forced configuration."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
shapes = json.load(open(os.path.join(ROOT, 'gpurun_out', 'gemm_shapes.json')))
NAMES = {1: '128x128', 2: '128x64', 3: '64x128', 4: '64x64', 5: '128x32', 6: '32x128'}


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


rng = np.random.RandomState(0)
total_auto = total_best = 0.0
for d in shapes:
    M, N, Kd, tA, tB, batch, ragged = d['M'], d['N'], d['K'], d['tA'], d['tB'], d['batch'], d['ragged']
    if d['extra'] or (N >= 1140 and (M >= 1140 or ragged == 1) and max(Kd, d['max_ragged'] if ragged == 2 else 0) > 160):
        continue                                   # the dominant six (and Linear-over-cat forms): not part of this sweep
    gptr = None
    if ragged:
        tot, mx = d['ragged_total'], d['max_ragged']
        if batch in (32,):
            counts = rng.randint(int(0.8 * tot / batch), int(1.2 * tot / batch) + 1, size=batch)
            counts = (counts * (tot / counts.sum())).astype(int)
        else:
            counts = np.full(batch, mx)
            counts[-1] = max(tot - mx * (batch - 1), 1)
        counts = np.maximum(counts, 1)
        mx = int(counts.max())
        gp = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
        n = int(counts.sum())
        if ragged == 2:
            A, Bm, C = torch.randn(n, M, device=dev), torch.randn(n, N, device=dev), torch.empty(batch, M, N, device=dev)
            call = lambda: K.gemm(A, Bm, C, M, N, 0, True, False, M, N, N, 1.0, 0.0, None, batch, 0, 0, M * N, gp, 2, mx, n)
        else:
            A = torch.randn(n, Kd, device=dev)
            Bm = torch.randn(batch, N, Kd, device=dev) if tB else torch.randn(batch, Kd, N, device=dev)
            C = torch.empty(n, N, device=dev)
            call = lambda: K.gemm(A, Bm, C, 0, N, Kd, False, bool(tB), Kd, Bm.shape[2], N, 1.0, 0.0, None, batch, 0, Kd * N, 0, gp, 1, mx, n)
        flops = 2.0 * n * N * (M if ragged == 2 else Kd)
    else:
        A = torch.randn(batch, *((Kd, M) if tA else (M, Kd)), device=dev)
        Bm = torch.randn(batch, *((N, Kd) if tB else (Kd, N)), device=dev)
        C = torch.empty(batch, M, N, device=dev)
        call = lambda: K.gemm(A, Bm, C, M, N, Kd, bool(tA), bool(tB), A.shape[2], Bm.shape[2], N, 1.0, 0.0, None, batch,
                              A.shape[1] * A.shape[2], Bm.shape[1] * Bm.shape[2], M * N)
        flops = 2.0 * batch * M * N * Kd
    K.lib.cgc_gemm_tuning(0)
    t_auto = timeit(call)
    res = {}
    kext = d['max_ragged'] if ragged == 2 else Kd
    for cfg in range(1, 7):
        for mode in ((10, 20) if kext <= 256 else (10,)):
            K.lib.cgc_gemm_tuning(cfg + mode)
            res[(cfg, mode)] = timeit(call, 5)
    K.lib.cgc_gemm_tuning(0)
    best = min(res, key=res.get)
    total_auto += t_auto * d['count']
    total_best += min(t_auto, res[best]) * d['count']
    print('M=%6d N=%5d K=%5d %s%s b=%3d rag=%d x%d  auto %7.1f us (%5.1f TF) | best %s%s %7.1f us (%5.1f TF)  | %s' % (
        M, N, kext, 'T' if tA else 'N', 'T' if tB else 'N', batch, ragged, d['count'], t_auto, flops / t_auto / 1e6,
        NAMES[best[0]], 'p' if best[1] == 10 else 's', res[best], flops / res[best] / 1e6,
        ' '.join('%s%s:%.0f' % (NAMES[c], 'p' if m == 10 else 's', v) for (c, m), v in sorted(res.items()))), flush=True)
print('sum over the step: automatic %.0f us, best-per-shape %.0f us' % (total_auto, total_best))
