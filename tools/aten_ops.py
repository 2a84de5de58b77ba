#!/usr/bin/env python
"""Which aten ops (torch glue around the C-ABI launches) a training step issues: counts per step by op and by call site."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
B = 4
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                      norm_adj=True, jk=True, drop_out=0.2).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
torch.autograd.set_multithreading_enabled(False)


def step():
    _, loss = model(b)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
counts, sites = collections.Counter(), collections.Counter()
SKIP = ('aten.empty', 'aten.view', 'aten.as_strided', 'aten.slice.', 'aten.select.', 'aten.detach', 'aten.t.', 'aten.transpose',
        'aten.reshape', 'aten._unsafe_view', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.alias', 'aten.permute', 'aten.stride',
        'aten.sym_', 'aten.is_', 'aten.size', 'aten.numel', 'aten.dim', 'aten.unbind', 'aten.split', 'aten._local_scalar')


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            counts[name] += 1
            st = [f for f in traceback.extract_stack() if 'cgc-net_amd' in f.filename]
            if st:
                f = st[-1]
                sites['%s  <- %s:%d %s' % (name, os.path.basename(f.filename), f.lineno, f.name)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step()
torch.cuda.synchronize()
print('aten ops launching work, one step:', sum(counts.values()))
for k, v in counts.most_common(30):
    print('%4d  %s' % (v, k))
print('--- by call site')
for k, v in sites.most_common(60):
    print('%4d  %s' % (v, k))
