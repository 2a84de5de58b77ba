#!/usr/bin/env python
"""Every cgc_gemm_f32 call of one training step (C3 default workload): shape, layout, batch, raggedness, extra segments -- the
work list for the tile-shape sweep (tools/gemm_cfg_sweep.py).  Writes gpurun_out/gemm_shapes.json."""
import collections
import json
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
calls = collections.Counter()
orig = K.gemm


def spy(A, Bm, C, M, N, Kd, tA, tB, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None, batch=1, sA=0, sB=0, sC=0, gptr=None,
        ragged=0, max_ragged=0, ragged_total=0, extra=()):
    calls[(M, N, Kd, int(tA), int(tB), batch, ragged, max_ragged, ragged_total, tuple(e[4] for e in extra), float(beta) != 0.0)] += 1
    return orig(A, Bm, C, M, N, Kd, tA, tB, lda, ldb, ldc, alpha, beta, bias, batch, sA, sB, sC, gptr, ragged, max_ragged, ragged_total, extra)


_, loss = model(b)
loss.backward()
K.gemm = spy
model.zero_grad()
_, loss = model(b)
loss.backward()
torch.cuda.synchronize()
K.gemm = orig
out = [dict(M=k[0], N=k[1], K=k[2], tA=k[3], tB=k[4], batch=k[5], ragged=k[6], max_ragged=k[7], ragged_total=k[8], extra=list(k[9]),
            beta=k[10], count=v) for k, v in calls.items()]
out.sort(key=lambda d: -(2.0 * (d['ragged_total'] if d['ragged'] else d['M'] * d['batch']) * d['N'] * max(d['K'], d['max_ragged'] if d['ragged'] == 2 else 0)))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'gemm_shapes.json'), 'w'), indent=0)
for d in out:
    print(d)
