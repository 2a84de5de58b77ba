#!/usr/bin/env python
"""Per-call GPU time of every cgc_gemm_f32 call inside one training step (C3 default workload, shipped flags): events around
each call, median over a few steps.  Shows which products are far from what their operand traffic allows."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels, network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                      norm_adj=True, jk=True, drop_out=0.2).to(dev)
K = kernels.get()
times = collections.defaultdict(list)
orig = K.gemm


def spy(A, Bm, C, M, N, Kd, tA, tB, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None, batch=1, sA=0, sB=0, sC=0, gptr=None,
        ragged=0, max_ragged=0, ragged_total=0, extra=()):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(A, Bm, C, M, N, Kd, tA, tB, lda, ldb, ldc, alpha, beta, bias, batch, sA, sB, sC, gptr, ragged, max_ragged, ragged_total, extra)
    e1.record()
    key = (M, N, Kd, 'T' if tA else 'N', 'T' if tB else 'N', batch, ragged, max_ragged, ragged_total, tuple(e[4] for e in extra), float(beta) != 0.0)
    times[key].append((e0, e1))
    return r


for _ in range(3):
    model.zero_grad()
    _, loss = model(b)
    loss.backward()
K.gemm = spy
for _ in range(5):
    model.zero_grad()
    _, loss = model(b)
    loss.backward()
torch.cuda.synchronize()
K.gemm = orig
rows = []
for k, ev in times.items():
    t = sorted(a.elapsed_time(c) * 1e3 for a, c in ev)
    per_step = len(ev) // 5
    rows.append((t[len(t) // 2] * per_step, t[len(t) // 2], per_step, k))
rows.sort(reverse=True)
print('%9s %9s %5s  %s' % ('us/step', 'us/call', 'calls', '(M, N, K, tA, tB, batch, ragged, max_ragged, ragged_total, extra K, beta)'))
for tot, med, n, k in rows:
    print('%9.1f %9.1f %5d  %s' % (tot, med, n, k))
print('total %.1f us per step' % sum(r[0] for r in rows))
