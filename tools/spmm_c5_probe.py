#!/usr/bin/env python
"""Wide aggregation A*S at the stress configuration (C5: ~8000-node graphs, width 1600): nodes in draw order vs listed grid
cell by grid cell (data.spatial_order).  Reports time and the fraction of the 8 TB/s HBM peak for the algorithmic bytes."""
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
B, N, W = int(os.environ.get("PROBE_B", "32")), int(sys.argv[1]) if len(sys.argv) > 1 else 8000, int(sys.argv[2]) if len(sys.argv) > 2 else 1600
for spatial in (False, True):
    ds = SyntheticCellGraphs(B, N, 64, base_seed=0, spatial=spatial)
    b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
    g = BatchGraph.from_batch(b, 0.4)
    n, nnz = g.n, g.nnz
    ld = -(-W // 32) * 32
    x = torch.randn(n, ld, device=dev)[:, :W]
    out = torch.empty(n, ld, device=dev)[:, :W]
    for name, args in (('A x', (g.rowptr, g.col, None, g.val)), ('A^T x', (g.t_rowptr, g.t_col, None, g.t_val))):
        def run():
            K.spmm(args[0], args[1], args[2], args[3], None, None, x, out, n, W, g.gptr, g.B, g.nmax, 0, ld)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        by = 8.0 * n * W + 4.0 * (n + 1) + 8.0 * nnz
        print('%-6s nodes/graph %d W %d %-13s %8.1f us  %6.1f GB/s = %.3f of 8 TB/s' % (
            name, N, W, 'spatial order' if spatial else 'draw order', ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 8000.0))
