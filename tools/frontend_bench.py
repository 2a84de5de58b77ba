#!/usr/bin/env python
"""Loader front-end for one C3 batch (32 graphs x ~1800 nuclei): host collate + per-tensor copies (+ host k-NN graph
construction) versus the device front-end (one packed copy + collate kernel + radius k-NN on the GPU)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd.data import Batch, Data, SyntheticCellGraphs, radius_graph  # noqa: E402

dev = 'cuda:0'
ds = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
items = [ds[i] for i in range(32)]
bare = [Data(x=d.x, pos=d.pos, y=d.y) for d in items]
mean, std = torch.zeros(16), torch.ones(16)


def timeit(fn, reps=15):
    """median and worst wall-clock per call (each call synchronised: a batch is usable when the call returns)"""
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        ts.append((time.time() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[-1]


def host_all():
    its = [Data(x=(d.x - mean) / std, pos=d.pos, y=d.y, edge_index=radius_graph(d.pos, 100.0, None, True, 8)) for d in bare]
    return Batch.from_data_list(its).to(dev)


print('edges given : host collate + .to(device)             %7.2f ms (worst %.1f)' % timeit(lambda: Batch.from_data_list(items).to(dev)))
print('edges given : device collate (1 copy + 1 kernel)     %7.2f ms (worst %.1f)' % timeit(lambda: Batch.from_data_list(items, device=dev, mean=mean, std=std)))
print('from pos    : host z-score + cKDTree + collate + copy %7.2f ms (worst %.1f)' % timeit(host_all, reps=3))
print('from pos    : device collate + z-score + radius k-NN  %7.2f ms (worst %.1f)' % timeit(lambda: Batch.from_data_list(bare, device=dev, knn=(100.0, 8), mean=mean, std=std)))
