#!/usr/bin/env python
"""cgc_sage_wide_fwd alone at the C3 shape ([57.7k, 20] -> [57.7k, 1140], statistics on): 20 launches back to back per sample, median of 5 samples (kernel + statistics finalize).  CGC_LIB selects a
variant library (tools/variant_lib.sh).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
n, Kin, F, ld = 57711, 20, 1140, 1152
torch.manual_seed(0)
agg = torch.randn(n, 40, device=dev)
W = torch.randn(Kin, F, device=dev) * 0.2
b = torch.randn(F, device=dev) * 0.1
hn = torch.empty(n, ld, device=dev)[:, :F]
rinv = torch.empty(n, device=dev)
rm, rv = torch.zeros(F, device=dev), torch.ones(F, device=dev)
nbt = torch.zeros((), dtype=torch.int64, device=dev)
mean, istd = torch.empty(F, device=dev), torch.empty(F, device=dev)
def call():
    return K.sage_wide_fwd(agg, 40, W, b, n, Kin, F, True, 1, hn, rinv, True, float(n), 1e-5, 0.1, rm, rv, nbt, mean, istd)


for _ in range(10):
    ok = call()
assert ok
ts = []
for rep in range(5):             # 20 calls back to back between two events: the host's ~50 us per call hide behind the GPU's work
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        call()
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3 / 20)
ts = sorted(ts)
h = (agg[:, :Kin] @ W + b)
h = h / h.norm(dim=1, keepdim=True).clamp_min(1e-12)
o = torch.relu(h).double()
m_ref, v_ref = o.mean(0), o.var(0, unbiased=False)
print('%-22s median %.1f us (min %.1f)  mean err %.1e  istd rel err %.1e' % (
    os.path.basename(kernels.lib_path()), ts[len(ts) // 2], ts[0], float((mean.double() - m_ref).abs().max()),
    float((istd.double() * torch.sqrt(v_ref + 1e-5) - 1).abs().max())))
