#!/usr/bin/env python
"""The ragged batched contractions of the DiffPool backward at C3 sizes, with and without the accumulate (beta = 1) and the
extra K segment:  dS (+)= P dA'^T (+ X dX'^T)   [NT, ragged M]   and   dP = S dA'   [NN, ragged M]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.graph import BatchGraph  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
ds_ = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
g = BatchGraph.from_batch(Batch.from_data_list([ds_[i] for i in range(32)]).to(dev))
n, c, dx = g.n, 1140, 60
p, s = torch.randn(n, c, device=dev), torch.randn(n, c, device=dev)
dao, dxo = torch.randn(g.B, c, c, device=dev), torch.randn(g.B, c, dx, device=dev)
embed = torch.randn(n, dx, device=dev)
out = torch.zeros(n, c, device=dev)


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


fl = 2.0 * n * c * c
for name, tB, beta, extra in (('NN ragged-M  S dA\'            beta 0', False, 0.0, ()),
                              ('NT ragged-M  P dA\'^T          beta 0', True, 0.0, ()),
                              ('NT ragged-M  P dA\'^T          beta 1', True, 1.0, ()),
                              ('NT ragged-M  P dA\'^T + X dX\'^T beta 0', True, 0.0, [(embed, dxo, dx, dx, dx, 0, c * dx)]),
                              ('NT ragged-M  P dA\'^T + X dX\'^T beta 1', True, 1.0, [(embed, dxo, dx, dx, dx, 0, c * dx)])):
    ms = timeit(lambda: K.gemm(p, dao, out, 0, c, c, False, tB, c, c, c, 1.0, beta, None, g.B, 0, c * c, 0, g.gptr, 1, g.nmax, n,
                               extra=extra))
    f = fl + (2.0 * n * dx * c if extra else 0.0)
    print('%-44s %7.3f ms  %6.1f TFLOP/s' % (name, ms, f / ms / 1e9))
