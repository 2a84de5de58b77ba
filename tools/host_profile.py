#!/usr/bin/env python
"""cProfile of the host side of training steps in the launch-bound regime (C3 graphs, max_num_nodes = 1800 => C1 = 180)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 32
MAXN = int(sys.argv[sys.argv.index('--maxn') + 1]) if '--maxn' in sys.argv else 1800
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
kw = dict(concat=True, load_data_sparse=True)
if '--shipped' in sys.argv:
    kw.update(norm_adj=True, jk=True, drop_out=0.2)
model = network.SoftPoolingGcnEncoder(MAXN, 16, 20, 20, True, True, 20, 3, 0.1, [50], **kw).to(dev)
from cgc_net_amd.optim import Adam  # noqa: E402
opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, model=model)      # as bench.py
torch.autograd.set_multithreading_enabled(False)                                       # backward on this thread: profiled too


def step():
    _, loss = model(b)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(40)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host issue time per step %.3f ms, wall per step %.3f ms' % ((t1 - t0) * 50, (t2 - t0) * 50))
