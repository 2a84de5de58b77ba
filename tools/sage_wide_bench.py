#!/usr/bin/env python
"""Time cgc_sage_wide_fwd (with statistics) at one shape and check it against the fp64 product.
usage: sage_wide_bench.py [n K F reps]   env: CGC_SAGE_WIDE_COLS=0|1, CGC_SAGE_WIDE_CHUNKS=..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 57696
Kin = int(sys.argv[2]) if len(sys.argv) > 2 else 20
F = int(sys.argv[3]) if len(sys.argv) > 3 else 1140
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
dev = 'cuda:0'
K = kernels.get()
g = torch.Generator().manual_seed(0)
agg = torch.randn(n, Kin, generator=g).to(dev)
W = (torch.randn(Kin, F, generator=g) * 0.2).to(dev)
b = (torch.randn(F, generator=g) * 0.1).to(dev)
hn, rinv = torch.empty(n, F, device=dev), torch.empty(n, device=dev)
rm, rv = torch.zeros(F, device=dev), torch.ones(F, device=dev)
nbt = torch.zeros((), dtype=torch.int64, device=dev)
mean, istd = torch.zeros(F, device=dev), torch.zeros(F, device=dev)


def run():
    assert K.sage_wide_fwd(agg, Kin, W, b, n, Kin, F, True, 1, hn, rinv, True, float(n), 1e-5, 0.1, rm, rv, nbt, mean, istd)


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
e0.record()
for _ in range(reps):
    hn.fill_(1.0)
e1.record()
torch.cuda.synchronize()
fill_us = e0.elapsed_time(e1) * 1e3 / reps
run()
torch.cuda.synchronize()
h = agg.double() @ W.double() + b.double()
nr = h.norm(dim=1, keepdim=True).clamp_min(1e-12)
ref = h / nr
o = torch.relu(ref)
err = float((hn.double() - ref).abs().max())
print('n %d K %d F %d: %.1f us per call [fill_: %.1f us] (%.2f TB/s of hn written)  max |hn - fp64| %.2e  rinv rel %.2e  mean %.2e  istd rel %.2e  [cols=%s chunks=%s]' % (
    n, Kin, F, us, fill_us, n * F * 4 / us / 1e6, err, float((rinv.double() * nr[:, 0] - 1).abs().max()),
    float((mean.double() - o.mean(0)).abs().max()), float((istd.double() * torch.sqrt(o.var(0, unbiased=False) + 1e-5) - 1).abs().max()),
    os.environ.get('CGC_SAGE_WIDE_COLS', '1'), os.environ.get('CGC_SAGE_WIDE_CHUNKS', '256')))
