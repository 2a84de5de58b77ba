#!/usr/bin/env python
"""Host issue time per training step split into forward / backward / optimizer (no GPU sync inside the step), for the
launch-bound regime.  usage: host_sections.py [--batch B] [--maxn N] [--fused-adam] [--st-autograd]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


dev = 'cuda:0'
B, MAXN = arg('--batch', 4), arg('--maxn', 11404)
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.2)
model = network.SoftPoolingGcnEncoder(MAXN, 16, 20, 20, True, True, 20, 3, 0.1, [50], **kw).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused='--fused-adam' in sys.argv)
if '--st-autograd' in sys.argv:
    torch.autograd.set_multithreading_enabled(False)
tf = tb = to = 0.0


def step(timed):
    global tf, tb, to
    t0 = time.perf_counter()
    _, loss = model(b)
    t1 = time.perf_counter()
    opt.zero_grad()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if timed:
        tf, tb, to = tf + t1 - t0, tb + t2 - t1, to + t3 - t2


for _ in range(5):
    step(False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('batch %d maxn %d %s: host fwd %.2f  bwd %.2f  opt %.2f  = %.2f ms/step issue, %.2f ms/step wall' % (
    B, MAXN, ' '.join(a for a in sys.argv[1:] if a.startswith('--f') or a.startswith('--s')), tf / N * 1e3, tb / N * 1e3, to / N * 1e3,
    (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
