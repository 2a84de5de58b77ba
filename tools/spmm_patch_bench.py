#!/usr/bin/env python
"""K4 (A S, width 1140 on 1152-float rows) stand-alone: the gather kernel (k_spmm_wide) against the LDS-staged neighbour-union kernel
(k_spmm_patch, visit bit 2) on graphs whose nodes are listed in draw order / grid cell by grid cell.  Forward graph and transpose.
usage: python tools/spmm_patch_bench.py [graphs] [mean nodes] [width] [features-ignored]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.graph import BatchGraph  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 1800
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1140
ld = (W + 31) // 32 * 32
for spatial in (False, True):
    ds = SyntheticCellGraphs(B, nodes, 4, base_seed=0, spatial=spatial)
    b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
    g = BatchGraph.from_batch(b, 0.4)
    n = g.n
    x = torch.randn(n, ld, device=dev)
    out = torch.empty(n, ld, device=dev)
    nnz = int(g.rowptr[n])
    by = 8.0 * n * W + 4.0 * (n + 1) + 8.0 * nnz
    for name, rp, col, val in (('A S', g.rowptr, g.col, g.val), ('A^T dP', g.t_rowptr, g.t_col, g.t_val)):
        for visit in (1, 5):
            if visit == 5 and not spatial:
                continue
            fn = lambda: K.spmm(rp, col, None, val, None, None, x[:, :W], out[:, :W], n, W, g.gptr, g.B, g.nmax, visit, ld, g.gorder)
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 20
            print('%-8s %-10s %-28s %7.1f us  %6.0f GB/s  (%.3f of 8 TB/s)   %d rows, %.1f MB algorithmic' % (
                name, 'grid cells' if spatial else 'draw', 'k_spmm_patch (LDS unions)' if visit == 5 else 'k_spmm_wide (L2 gathers)',
                ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 8000.0, n, by / 1e6))
