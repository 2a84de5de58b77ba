"""Do the wide BatchNorm row kernels care whether the saved pre-activation hn sits on 128-byte-aligned rows?  hn is [n, F] dense
(row stride F) in their ABI: F = 1140 (4560-byte rows, what the step uses) against F = 1152 (aligned rows, same traffic + 1 %)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgc_net_amd import kernels

K = kernels.get()
dev = 'cuda:0'
n = 57711


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for F in (1140, 1152):
    hn = torch.randn(n, F, device=dev)
    hn = hn / hn.norm(dim=1, keepdim=True)
    rinv = torch.rand(n, device=dev) + 0.5
    dy = torch.randn(n, 1152, device=dev)
    y = torch.empty(n, 1152, device=dev)
    dh = torch.empty(n, F, device=dev)
    mean, istd, gamma, beta = (torch.randn(F, device=dev) * 0.1, torch.rand(F, device=dev) + 0.5, torch.rand(F, device=dev) + 0.5,
                               torch.randn(F, device=dev))
    sums = torch.empty(2, F, device=dev)
    db = torch.empty(F, device=dev)
    big = torch.empty(64 * 1024 * 1024, device=dev)          # 256 MB: flush the Infinity Cache between timings? (kept simple: not used)
    t1 = bench(lambda: K.bn_act_apply(hn, n, F, 0, mean, istd, gamma, beta, y, 1152))
    t2 = bench(lambda: K.bn_bwd_reduce(dy, 1152, hn, n, F, 0, mean, istd, sums))
    t3 = bench(lambda: K.bn_act_l2_bwd(dy, 1152, hn, rinv, n, F, 0, True, 2, mean, istd, gamma, sums, float(n), dh, db))
    print('F = %d: apply %.1f us, backward reduce %.1f us, backward %.1f us' % (F, t1, t2, t3))
