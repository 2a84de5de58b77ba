#!/usr/bin/env python
"""Cell-graph construction (radius k-NN, r = 100 px, k = 8 + self) for one C3 batch: HIP kernels vs the host tree."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402
from cgc_net_amd.data import radius_graph  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
for B, nodes in ((32, 1800), (32, 8000)):
    rng = np.random.RandomState(0)
    counts = rng.randint(int(0.8 * nodes), int(1.2 * nodes) + 1, size=B)
    pos = [rng.uniform(0, np.sqrt(c * 1784.0), size=(c, 2)).astype(np.float32) for c in counts]
    t0 = time.time()
    host = [radius_graph(torch.from_numpy(p), 100.0, None, True, 8) for p in pos]
    t_host = time.time() - t0
    allpos = torch.from_numpy(np.concatenate(pos)).to(dev)
    gptr = torch.tensor(np.cumsum([0] + list(counts)), dtype=torch.int32, device=dev)
    for _ in range(3):
        ei = K.radius_knn(allpos, gptr, B, 100.0, 8, True)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        ei = K.radius_knn(allpos, gptr, B, 100.0, 8, True)
    torch.cuda.synchronize()
    t_dev = (time.time() - t0) / 10
    nnz = sum(h.shape[1] for h in host)
    assert ei.shape[1] == nnz
    print('%d graphs x ~%d nodes (%d nodes, %d edges): host cKDTree %.1f ms (1 thread), HIP %.3f ms incl. the edge-count sync '
          '(%.0fx; %.1f M nodes/s)' % (B, nodes, allpos.shape[0], nnz, t_host * 1e3, t_dev * 1e3, t_host / t_dev,
                                      allpos.shape[0] / t_dev / 1e6))
