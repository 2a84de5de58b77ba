#!/usr/bin/env python
"""'fuse' node sampling (70 % farthest-point + 30 % random, half of the nuclei) for a batch of 32 images: HIP vs numpy."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd.data import fuse_sample, sample_nodes_batch  # noqa: E402

dev = 'cuda:0'
for nodes in (3600, 16000):
    rng = np.random.RandomState(0)
    counts = [int(c) for c in rng.randint(int(0.8 * nodes), min(int(1.2 * nodes), 16384) + 1, size=32)]
    pos = [rng.uniform(0, np.sqrt(c * 1784.0), size=(c, 2)).astype(np.float32) for c in counts]
    t0 = time.time()
    for p_, c in zip(pos[:4], counts[:4]):
        fuse_sample(p_.astype(np.float64), c // 2, rng)
    t_host = (time.time() - t0) * 8                      # 4 of 32 graphs timed
    allpos = torch.from_numpy(np.concatenate(pos)).to(dev)
    sample_nodes_batch(allpos, counts, 0.5, 'fuse')
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        idx, ks = sample_nodes_batch(allpos, counts, 0.5, 'fuse')
    torch.cuda.synchronize()
    t_dev = (time.time() - t0) / 3
    print('32 images x ~%d nuclei -> %d kept: numpy loop %.0f ms (extrapolated from 4 images), HIP %.1f ms (%.0fx)'
          % (nodes, sum(ks), t_host * 1e3, t_dev * 1e3, t_host / t_dev))
