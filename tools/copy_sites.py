#!/usr/bin/env python
"""Where ops.py makes contiguous copies of strided tensors during one training step (each is a ~5 us copy kernel)."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network, ops  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
ds = SyntheticCellGraphs(8, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(8)]).to(dev)
model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                      norm_adj=True, jk=True, drop_out=0.2).to(dev)
torch.autograd.set_multithreading_enabled(False)
sites = collections.Counter()
orig = ops._f32c


def spy(t):
    if not (t.dtype == torch.float32 and t.is_contiguous()):
        st = [f for f in traceback.extract_stack()[:-1] if 'cgc-net_amd' in f.filename]
        f = st[-1]
        sites['%s:%d %s  shape %s strides %s' % (os.path.basename(f.filename), f.lineno, f.name, tuple(t.shape), t.stride())] += 1
    return orig(t)


ops._f32c = spy
_, loss = model(b)
loss.backward()
torch.cuda.synchronize()
for k, v in sites.most_common(40):
    print(v, k)
print('total', sum(sites.values()))
