#!/usr/bin/env python
"""Parameter-gradient accuracy of the DenseJK kernels alone at the benchmarked row count (57.7 k rows, C = 20): HIP and torch
fp32 on the CPU, each against an fp64 evaluation, max|a-b| / max|b| per parameter.  Inputs are shaped like the block outputs
(post-BatchNorm, unit variance); the upstream gradient is ~1/n like a mean loss."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402

n = int(sys.argv[sys.argv.index('--rows') + 1]) if '--rows' in sys.argv else 57728
C = 20
torch.manual_seed(0)
mod = network.DenseJK('lstm', C, 3)
g = torch.Generator().manual_seed(1)
xs = torch.randn(n, 3 * C, generator=g)
dout = torch.randn(n, C, generator=g) / n


def run(m, x, dy):
    x = x.clone().requires_grad_(True)
    seq = x.reshape(-1, 3, C)
    alpha, _ = m.lstm(seq)
    alpha = torch.softmax(m.att(alpha).squeeze(-1), dim=-1)
    y = (seq * alpha.unsqueeze(-1)).sum(dim=1)
    y.backward(dy)
    return y.detach(), x.grad, {k: p.grad.clone() for k, p in m.named_parameters()}


import copy
m32, m64 = copy.deepcopy(mod), copy.deepcopy(mod).double()
y32, dx32, g32 = run(m32, xs, dout)
y64, dx64, g64 = run(m64, xs.double(), dout.double())
hip = copy.deepcopy(mod).to('cuda:0')
xh = xs.to('cuda:0').requires_grad_(True)
yh = hip(xh)
yh.backward(dout.to('cuda:0'))
gh = {k: p.grad.cpu() for k, p in hip.named_parameters()}


def strict(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


print('rows', n, 'lib', os.environ.get('CGC_LIB', 'default'))
print('%-28s %10s %10s' % ('', 'hip', 'torch fp32'))
print('%-28s %10.2e %10.2e' % ('out', strict(yh, y64), strict(y32, y64)))
print('%-28s %10.2e %10.2e' % ('dx', strict(xh.grad, dx64), strict(dx32, dx64)))
for k in g64:
    if float(g64[k].abs().max()) < 1e-14:
        continue
    print('%-28s %10.2e %10.2e' % (k, strict(gh[k], g64[k]), strict(g32[k], g64[k])))
