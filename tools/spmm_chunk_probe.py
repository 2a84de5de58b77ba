#!/usr/bin/env python
"""Probe: does the wide aggregation run faster when it follows its producer chunk by chunk (a few graphs at a time), so
that the gathered rows are still in L2 / Infinity Cache?  Times only the SpMM launches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.graph import BatchGraph  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
ds = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(32)]).to(dev)
g = BatchGraph.from_batch(b)
n, W = g.n, 1140
src = torch.randn(n, W, device=dev)
x = torch.empty_like(src)
out = torch.empty_like(src)
other = torch.empty(2, n, W, device=dev)
gp = g.gptr_host if isinstance(g.gptr_host, list) else list(g.gptr_host)
by = 8.0 * n * W + 4.0 * (n + 1) + 4.0 * g.nnz
for transpose in (False, True):
    rp, cl = (g.t_rowptr, g.t_col) if transpose else (g.rowptr, g.col)
    for G in (32, 16, 8, 4, 2, 1):
        tot = []
        for it in range(5):
            other.fill_(0.5)        # whatever ran before: push x / out out of the caches
            evs = []
            for g0 in range(0, 32, G):
                r0, r1 = int(gp[g0]), int(gp[min(g0 + G, 32)])
                torch.softmax(src[r0:r1], -1, out=x[r0:r1])          # producer of this chunk
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                K.spmm(rp[r0:r1 + 1], cl, None, None, None, None, x, out[r0:r1], r1 - r0, W)
                e.record()
                evs.append((s, e))
            torch.cuda.synchronize()
            tot.append(sum(s.elapsed_time(e) for s, e in evs))
        ms = float(np.median(tot[1:]))
        print('%s chunk=%2d graphs: SpMM total %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)'
              % ('A^T' if transpose else 'A  ', G, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80))
