#!/usr/bin/env python
"""Where does a gradient difference enter?  Taps d(loss)/d(layer output) of every convolution layer (after activation +
BatchNorm) and d(loss)/d(DenseJK input) in the HIP path, the fp32 oracle and the fp64 oracle, and prints max|a-b|/max|b|
against fp64 per tap, last level first.  The fp64 evaluation uses the HIP path's discrete decisions at the points fp32
cannot decide (tests/discrete.py) unless --natural is given.

    python tools/grad_taps.py [--batch 32] [--seed 0] [--plain|--jk-only] [--natural] [--detail GCN_embed_2.gcn1]

--detail: the rows of that tap with the largest error, and the pre-activations of the NEXT layer at the worst row (a value at
fp32 resolution of zero there = a ReLU whose sign two fp32 evaluations decide differently)."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cgc_net_amd  # noqa: E402,F401
import discrete  # noqa: E402
from cgc_net_amd import network, ops  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402
from oracle import dense_ref  # noqa: E402


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


B, seed = arg('--batch', 32), arg('--seed', 0)
flags = dict(jk=True) if '--jk-only' in sys.argv else (dict() if '--plain' in sys.argv else dict(norm_adj=True, jk=True))
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=seed)
cpu_batch = Batch.from_data_list([ds[i] for i in range(B)])
args = (11404, 16, 20, 20, True, True, 20, 3, 0.1, [50])
kw = dict(concat=True, gcn_name='SAGE', load_data_sparse=True, drop_out=0.)
kw.update(flags)
torch.manual_seed(0)
ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw)
model = network.SoftPoolingGcnEncoder(*args, **kw)
model.load_state_dict(ref.state_dict())
model.to('cuda:0').train()
ref.train()
ref64 = copy.deepcopy(ref).double()
ref64.load_data_sparse = False
adj = dense_ref.to_dense_adj(cpu_batch.edge_index, cpu_batch.batch)
xd, counts_t = dense_ref.to_dense_batch(cpu_batch.x, cpu_batch.batch)
counts = [int(c) for c in counts_t]
inp64 = (xd.double(), adj.double(), counts_t, cpu_batch.y)

# ---- HIP: decisions recorded by tests/discrete.py, gradient taps on top of its wrapper
names = {id(p): k[:-len('.weight')] for k, p in model.named_parameters() if k.endswith('.weight')}
taps_h = {}
with discrete.record_hip_decisions(model) as dec:
    recorded_sage = ops.sage_project

    def sage_project_tap(agg, weight, *a, **k):
        h = recorded_sage(agg, weight, *a, **k)
        if h.requires_grad:
            h.register_hook(lambda g, name=names[id(weight)]: taps_h.__setitem__(name, g.detach().cpu()))
        return h
    ops.sage_project = sage_project_tap
    jk_forward = network.DenseJK.forward

    def jk_tap(self, xs):
        name = [k for k, m in model.named_modules() if m is self][0]
        if xs.requires_grad:
            xs.register_hook(lambda g, name=name: taps_h.__setitem__(name + '.in', g.detach().cpu()))
        return jk_forward(self, xs)
    network.DenseJK.forward = jk_tap
    try:
        _, loss = model(cpu_batch.to('cuda:0'))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        network.DenseJK.forward = jk_forward
        ops.sage_project = recorded_sage


# ---- oracle taps
def tapped(run, m):
    taps = {}
    mods = {id(mm): k for k, mm in m.named_modules()}
    bn0, jk0 = dense_ref.GNNBlock._bn, dense_ref.DenseJK.forward

    def bn_tap(self, k, h):
        out = bn0(self, k, h)
        out.register_hook(lambda g, name='%s.gcn%d' % (mods[id(self)], k): taps.__setitem__(name, g.detach()))
        return out

    def jk_tap(self, xs):
        xs.register_hook(lambda g, name=mods[id(self)] + '.in': taps.__setitem__(name, g.detach()))
        return jk0(self, xs)
    dense_ref.GNNBlock._bn, dense_ref.DenseJK.forward = bn_tap, jk_tap
    try:
        run()
    finally:
        dense_ref.GNNBlock._bn, dense_ref.DenseJK.forward = bn0, jk0
    return taps


def run32():
    _, l = ref(cpu_batch)
    l.backward()


t32 = tapped(run32, ref)
_, _, pre64, embeds64 = discrete.run_oracle_recording(ref64, inp64)
routing, masks, wf, rf = discrete.hip_choices(dec, pre64, embeds64, counts)
print('HIP decisions differing from the natural fp64 evaluation: %d readout winners, %d ReLU signs' % (wf, rf))
if '--natural' in sys.argv:
    routing, masks = [e.argmax(1) for e in embeds64], {k: v > 0 for k, v in pre64.items()}
t64 = tapped(lambda: discrete.run_oracle_routed(ref64, inp64, routing, masks), ref64)
_, _, pre32, _ = discrete.run_oracle_recording(ref, cpu_batch)


def strict(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def to_dense(flat, like):
    return discrete._to_dense(flat, counts, like)


real = torch.arange(max(counts)).unsqueeze(0) < torch.tensor(counts).unsqueeze(1)
print('%-28s %10s %10s   (max|d - d64| / max|d64| of d loss / d tap)' % ('tap', 'hip', 'fp32 oracle'))
for name in sorted(t64, key=lambda s: (-int(s.split('.')[0][-1]), s)):
    if name not in taps_h:
        print('%-28s   (not tapped in the HIP path)' % name)
        continue
    d64, d32 = t64[name], t32[name]
    dh = to_dense(taps_h[name], d64)
    if d64.shape[1] == max(counts):         # level 1: the rows behind a graph's nodes exist only in the dense layout
        d64, d32, dh = d64[real], d32[real], dh[real]
    print('%-28s %10.2e %10.2e' % (name, strict(dh, d64), strict(d32, d64)))
    if name == arg('--detail', ''):
        F = d64.shape[-1]
        scale = float(d64.abs().max())
        rows = ((dh.double() - d64).abs() / scale).reshape(-1, F).max(dim=1).values
        top = rows.topk(8)
        print('   scale %.3e; rows with error > 1e-5: %d of %d; > 1e-4: %d; worst rows %s' % (
            scale, int((rows > 1e-5).sum()), rows.numel(), int((rows > 1e-4).sum()),
            [(int(i), '%.1e' % float(v)) for v, i in zip(top.values, top.indices)]))
        i = int(top.indices[0])
        nxt = name[:-1] + str(int(name[-1]) + 1)
        if nxt in pre64 and pre64[nxt].shape[1] != max(counts):
            print('   pre-activations of %s at that row, fp64:       ' % nxt, pre64[nxt].reshape(-1, pre64[nxt].shape[-1])[i].tolist())
            print('   ... fp32 oracle:', pre32[nxt].reshape(-1, pre32[nxt].shape[-1])[i].tolist())
            print('   ... HIP:        ', dec.preact[nxt][i].tolist())
        print('   d tap at that row, HIP        ', dh.reshape(-1, F)[i].tolist())
        print('   d tap at that row, fp64       ', d64.reshape(-1, F)[i].tolist())
        print('   d tap at that row, fp32 oracle', d32.reshape(-1, F)[i].tolist())
