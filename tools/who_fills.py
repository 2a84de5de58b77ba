#!/usr/bin/env python
"""torch.profiler on two training steps: which host-side ops the small torch kernels (fill / copy / add / cat) come from."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
ds = SyntheticCellGraphs(8, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(8)]).to(dev)
model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                      norm_adj=True, jk=True, drop_out=0.2).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
torch.autograd.set_multithreading_enabled(False)


def step():
    _, loss = model(b)
    opt.zero_grad()
    torch.mean(loss).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::'):
        cnt[e.name] += 1
for k, v in cnt.most_common(45):
    print(v, k)
print('--- host ops that launched fill / copy / elementwise kernels')
src = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU:
        for k in getattr(e, 'kernels', []):
            if any(t in k.name for t in ('FillFunctor', 'fillBuffer', 'copyBuffer', 'direct_copy', 'CUDAFunctor_add', 'CatArray')):
                src[(e.name, k.name[:60])] += 1
for (op, kn), v in src.most_common(40):
    print(v, op, '->', kn)
