#!/usr/bin/env python
"""Are the gradient differences between two fp32 evaluations of the model discrete routing flips?  Records the max-readout
winners (graph, channel) -> row of the HIP path, of the fp32 oracle and of the fp64 oracle at every level, and the margin by
which each fp64 winner wins: a readout whose two best rows differ by less than fp32 rounding sends its gradient to a different
row in different evaluations -- same value, different gradient."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cgc_net_amd  # noqa: E402,F401
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

hip_embeds = []
orig = ops.segment_max


def spy(x, gptr, Bn, nmax):
    hip_embeds.append((x.detach().cpu(), gptr.cpu().tolist(), nmax))
    return orig(x, gptr, Bn, nmax)


ops.segment_max = spy
network.ops.segment_max = spy
model(cpu_batch.to('cuda:0'))
torch.cuda.synchronize()


def dense_embeds(m, inp):
    rec = []
    st = m._stage

    def stage(k, x, adj, mask):
        e, ro = st(k, x, adj, mask)
        rec.append(e.detach())
        return e, ro
    m._stage = stage
    m(inp)
    return rec


e32 = dense_embeds(ref, cpu_batch)
ref64.load_data_sparse = False
adj = dense_ref.to_dense_adj(cpu_batch.edge_index, cpu_batch.batch)
xd, counts = dense_ref.to_dense_batch(cpu_batch.x, cpu_batch.batch)
e64 = dense_embeds(ref64, (xd.double(), adj.double(), counts, cpu_batch.y))
for lvl in range(3):
    xh, gptr, nmax = hip_embeds[lvl]
    d32, d64 = e32[lvl], e64[lvl]
    Bn, N, D = d64.shape
    dense_h = torch.zeros(Bn, N, D)
    for b in range(Bn):
        dense_h[b, :gptr[b + 1] - gptr[b]] = xh[gptr[b]:gptr[b + 1]]
    a64, a32, ah = d64.argmax(1), d32.argmax(1), dense_h.argmax(1)
    top2 = d64.topk(2, dim=1).values
    margin = ((top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30))
    print('level %d: %d readouts; winners differing from fp64: hip %d, fp32 oracle %d; hip vs fp32 oracle %d; '
          'fp64 margins below 1e-6: %d, smallest %.1e; max |value| difference hip/fp32 vs fp64: %.1e / %.1e'
          % (lvl + 1, Bn * D, int((ah != a64).sum()), int((a32 != a64).sum()), int((ah != a32).sum()),
             int((margin < 1e-6).sum()), float(margin.min()),
             float((dense_h.double() - d64).abs().max()), float((d32.double() - d64).abs().max())))
