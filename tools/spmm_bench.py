#!/usr/bin/env python
"""Micro-benchmark of the wide aggregation SpMM (K4: A*S, width = cluster count) on C3-shaped graphs.  GPU only.
Reports algorithmic GB/s = (8 n W + 4(n+1) + 4 nnz) / time.  usage: python tools/spmm_bench.py [width ...]"""
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
widths = [int(a) for a in sys.argv[1:]] or [1140]
ds = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(32)]).to(dev)
g = BatchGraph.from_batch(b)
n, nnz = g.n, g.nnz
for W in widths:
    x = torch.randn(n, W, device=dev)
    out = torch.empty_like(x)
    def run():
        K.spmm(g.rowptr, g.col, None, None, None, None, x, out, n, W, g.gptr, g.B, g.nmax)
    def run_t():
        K.spmm(g.t_rowptr, g.t_col, None, None, None, None, x, out, n, W, g.gptr, g.B, g.nmax)
    gw = BatchGraph.from_batch(b, 0.4)                     # re-normalised adjacency: per-edge weights (shipped flags)
    tval = gw.val[gw.t_perm.long()].contiguous()            # weights already in transposed slot order

    def run_w():
        K.spmm(gw.rowptr, gw.col, None, gw.val, None, None, x, out, n, W, gw.gptr, gw.B, gw.nmax)

    def run_tw():
        K.spmm(gw.t_rowptr, gw.t_col, gw.t_perm, gw.val, None, None, x, out, n, W, gw.gptr, gw.B, gw.nmax)

    def run_tw2():
        K.spmm(gw.t_rowptr, gw.t_col, None, tval, None, None, x, out, n, W, gw.gptr, gw.B, gw.nmax)
    for name, fn in (('A x', run), ('A^T x', run_t), ('Aw x', run_w), ('Aw^T x (perm)', run_tw), ('Aw^T x (pre-permuted w)', run_tw2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        # in-step conditions: x has just been written by its producer (a row softmax), only the SpMM is timed
        src, tt = torch.randn(n, W, device=dev), []
        for _ in range(8):
            torch.softmax(src, -1, out=x)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            tt.append(s.elapsed_time(e))
        after = sorted(tt[2:])[len(tt[2:]) // 2]
        by = 8.0 * n * W + 4.0 * (n + 1) + 4.0 * nnz
        print('%-24s n=%d nnz=%d W=%5d  %8.1f us  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s); right after its producer '
              '%.1f us (%.1f%%)  [env %s]' % (
            name, n, nnz, W, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80.0, after * 1e3, by / after / 1e6 / 80.0,
            {k: v for k, v in os.environ.items() if k.startswith('CGC_SPMM')}))
