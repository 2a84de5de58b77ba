#!/usr/bin/env python
"""The same short training run -- C3-sized batches (8 graphs of ~1800 nodes: the six dominant products are above the fp16 mode's
threshold), the reference's optimiser (Adam, lr 1e-3, weight decay 1e-4), same seed, same batches -- in the three GEMM modes: the loss
per step and the largest relative parameter difference to the exact run at the end.  GPU only.  usage: python tools/train_modes.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.optim import Adam  # noqa: E402

dev = 'cuda:0'
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ds = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
batches = [Batch.from_data_list([ds[b * 8 + i] for i in range(8)]).to(dev) for b in range(4)]
runs = {}
for mode, name in ((0, 'exact'), (1, 'bf16 x 6'), (2, 'fp16 x 3')):
    torch.manual_seed(0)
    model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True, norm_adj=True,
                                          jk=True, drop_out=0.0).to(dev).train()
    model.gemm_mode = mode
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, model=model)
    losses = []
    for s in range(steps):
        _, loss = model(batches[s % 4])
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    runs[mode] = (losses, {k: p.detach().clone() for k, p in model.named_parameters()}, name)
ex = runs[0]
print('step  ' + '  '.join('%-12s' % runs[m][2] for m in (0, 1, 2)))
for s in range(steps):
    if s < 5 or s % 5 == 4:
        print('%4d  ' % s + '  '.join('%-12.6f' % runs[m][0][s] for m in (0, 1, 2)))
for m in (1, 2):
    dl = max(abs(a - b) / max(abs(a), 1e-9) for a, b in zip(ex[0], runs[m][0]))
    dp = max(float((runs[m][1][k] - ex[1][k]).abs().max()) / max(float(ex[1][k].abs().max()), 1e-9) for k in ex[1])
    print('%s vs exact after %d steps: largest relative loss difference %.2e, largest relative parameter difference (max-norm per tensor) %.2e' % (runs[m][2], steps, dl, dp))
