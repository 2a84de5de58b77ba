#!/usr/bin/env python
"""What a mixed read/write stream reaches on this part (the wide row kernels move 263 MB tensors at ~4.1 TB/s): torch's own
fill / copy / add / in-place scale on a [57696, 1140] fp32 tensor, next to cgc_bn_act_apply (1 read + 1 write) and the in-place softmax."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
n, F = 57696, 1140
x = torch.randn(n, F, device=dev)
y = torch.empty_like(x)
z = torch.empty_like(x)
MB = n * F * 4 / 1e6


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


K = kernels.get()
mean, istd, gamma, beta = (torch.randn(F, device=dev) for _ in range(4))
rows = [('fill_ (1 write)', lambda: y.fill_(1.0), 1),
        ('copy_ (1 read + 1 write)', lambda: y.copy_(x), 2),
        ('mul_ in place (1 read + 1 write, same lines)', lambda: y.mul_(1.0001), 2),
        ('torch.add out= (2 reads + 1 write)', lambda: torch.add(x, y, out=z), 3),
        ('sum (1 read)', lambda: x.sum(), 1),
        ('cgc_bn_act_apply (1 read + 1 write)', lambda: K.bn_act_apply(x, n, F, 1, mean, istd, gamma, beta, y, F), 2)]
for name, fn, passes in rows:
    us = timed(fn)
    print('%-50s %7.1f us  %5.2f TB/s' % (name, us, passes * MB / us))
