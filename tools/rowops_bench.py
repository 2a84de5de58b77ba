#!/usr/bin/env python
"""The streaming row kernels on the [57.7 k, 1140] tensors of the step, isolated: microseconds and TB/s of algorithmic traffic,
cold (a 1 GiB buffer is swept between launches so that nothing is left in the 256 MB Infinity Cache) and warm."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

K = kernels.get()
dev = 'cuda:0'
n, F = 57728, 1140
g = torch.Generator(device=dev).manual_seed(0)
dy, hn, out = (torch.randn(n, F, device=dev, generator=g) for _ in range(3))
rinv = torch.rand(n, device=dev) + 0.5
mean, istd, gamma, beta = (torch.rand(F, device=dev) + 0.5 for _ in range(4))
sums = torch.randn(2, F, device=dev)
db = torch.empty(F, device=dev)
flush = torch.empty(256 * 1024 * 1024, device=dev)
A = torch.rand(32, 1140, 1140, device=dev)
At, An = torch.empty_like(A), torch.empty_like(A)
invd, ge1 = torch.empty(32 * 1140, device=dev), torch.empty(32 * 1140, device=dev)
gAn, gAt, dA = torch.randn_like(A), torch.randn_like(A), torch.empty_like(A)
MB = n * F * 4 / 1e6
AMB = A.numel() * 4 / 1e6
CASES = [
    ('bn_act_apply        (R+W)', 2 * MB, lambda: K.bn_act_apply(hn, n, F, 1, mean, istd, gamma, beta, out, F)),
    ('bn_bwd_reduce       (2R)', 2 * MB, lambda: K.bn_bwd_reduce(dy, F, hn, n, F, 1, mean, istd, sums)),
    ('bn_act_l2_bwd       (2R+W)', 3 * MB, lambda: K.bn_act_l2_bwd(dy, F, hn, rinv, n, F, 1, True, 2, mean, istd, gamma, sums, float(n), out, db)),
    ('softmax_fwd         (R+W)', 2 * MB, lambda: K.softmax_fwd(dy, n, F, out)),
    ('softmax_bwd         (2R+W)', 3 * MB, lambda: K.softmax_bwd(hn, dy, n, F, out, db)),
    ('adj_prep_fwd        (R+2W)', 3 * AMB, lambda: K.adj_prep_fwd(A, 32 * 1140, 1140, 0.4, At, An, invd, ge1)),
    ('adj_prep_bwd        (4R+W)', 5 * AMB, lambda: K.adj_prep_bwd(A, An, invd, ge1, gAn, gAt, 32 * 1140, 1140, 0.4, dA)),
    ('torch copy          (R+W)', 2 * MB, lambda: out.copy_(dy)),
]


def timeit(fn, cold):
    ts = []
    for _ in range(6):
        if cold:
            flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2]


print('%-32s %10s %10s %10s %10s' % ('kernel', 'cold us', 'TB/s', 'warm us', 'TB/s'))
for name, mb, fn in CASES:
    tc, tw = timeit(fn, True), timeit(fn, False)
    print('%-32s %10.1f %10.2f %10.1f %10.2f' % (name, tc, mb / tc, tw, mb / tw))
