#!/usr/bin/env python
"""Per-call timing of every C-ABI launch in one training step of the C3 workload (HIP events around each call).
usage: python tools/step_breakdown.py [--maxn 11404] [--flags plain|shipped]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels, network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--maxn', type=int, default=11404)
ap.add_argument('--flags', default='plain')
ap.add_argument('--nodes', type=int, default=1800)
args = ap.parse_args()
dev = 'cuda:0'
K = kernels.get()
records = []


def wrap(name):
    fn = getattr(K, name)

    def inner(*a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **kw)
        e.record()
        if name == 'gemm':
            A, B, C, M, N, Kd, tA, tB = a[:8]
            batch = a[14] if len(a) > 14 else kw.get('batch', 1)
            ragged = a[19] if len(a) > 19 else kw.get('ragged', 0)
            tag = 'gemm %s%s M=%d N=%d K=%d b=%d r=%d' % ('T' if tA else 'N', 'T' if tB else 'N', M, N, Kd, batch, ragged)
        elif name == 'spmm':
            tag = 'spmm W=%d #%d' % (a[9], sum(1 for r_ in records if r_[0].startswith('spmm W=%d ' % a[9])))
        elif name in ('l2norm_act_stats', 'bn_act_apply', 'bn_bwd_reduce', 'bn_act_l2_bwd', 'colsum', 'softmax_fwd', 'softmax_bwd'):
            dims = [x for x in a if isinstance(x, int)][:3]
            tag = '%s %s' % (name, dims)
        else:
            tag = name
        records.append((tag, s, e))
        return r
    setattr(K, name, inner)


ds = SyntheticCellGraphs(32, args.nodes, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(32)]).to(dev)
kw = dict(concat=True, load_data_sparse=True)
if args.flags == 'shipped':
    kw.update(norm_adj=True, jk=True, drop_out=0.2)
model = network.SoftPoolingGcnEncoder(args.maxn, 16, 20, 20, True, True, 20, 3, 0.1, [50], **kw).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)


def step():
    _, loss = model(b)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
for n in [m for m in dir(kernels.KernelSpec) if not m.startswith('_')]:
    wrap(n)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
step()
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for tag, s, e in records:
    a = agg.setdefault(tag, [0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print('step %.3f ms ; C-ABI calls %d, %.3f ms inside call brackets' % (t0.elapsed_time(t1), len(records), tot))
for tag, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%-58s x%-3d %8.3f ms  (%6.1f us each)' % (tag, c, ms, 1e3 * ms / c))
