#!/usr/bin/env python
"""Probe: how much does a locality-preserving node order (Morton curve on the nucleus coordinates, per graph) help the
wide aggregation SpMM?  Compares the generator's random node order with a Morton order of the same graphs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402
from cgc_net_amd.data import Batch, Data, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.graph import BatchGraph  # noqa: E402

dev = 'cuda:0'
K = kernels.get()


def morton(pos, cell=64.0):
    q = np.floor(pos / cell).astype(np.uint32)
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x0000FFFF0000FFFF
        v = (v | (v << 8)) & 0x00FF00FF00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F
        v = (v | (v << 2)) & 0x3333333333333333
        v = (v | (v << 1)) & 0x5555555555555555
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1)


def reorder(d):
    perm = np.argsort(morton(d.pos.numpy()), kind='stable')
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    ei = torch.from_numpy(inv)[d.edge_index]
    return Data(x=d.x[perm], pos=d.pos[perm], y=d.y, edge_index=ei)


ds = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
graphs = [ds[i] for i in range(32)]
for name, gl in (('random order', graphs), ('morton order', [reorder(g) for g in graphs])):
    b = Batch.from_data_list(gl).to(dev)
    g = BatchGraph.from_batch(b)
    n = g.n
    for W in (1140,):
        x = torch.randn(n, W, device=dev)
        out = torch.empty_like(x)
        big = torch.empty(300 * 1024 * 1024 // 4, device=dev)     # flush: > Infinity Cache
        ts = []
        for it in range(6):
            big.fill_(1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            K.spmm(g.rowptr, g.col, None, None, None, None, x, out, n, W, g.gptr, g.B, g.nmax)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ms = float(np.median(ts[1:]))
        by = 8.0 * n * W + 4.0 * (n + 1) + 4.0 * g.nnz
        print('%-13s W=%d cache-cold: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)' % (name, W, ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80))
