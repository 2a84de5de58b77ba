"""Is the dominant GEMM's k loop held back by operand latency?  The same 4320 tiles of 36 k-tiles once with every batch item reading
the SAME 15 MB of operands (L2 / Infinity Cache resident) and once with distinct operands (430 MB: the step's situation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgc_net_amd import kernels
K = kernels.get()
dev = 'cuda:0'
M, N, Kd, LD, batch = 2048, 1140, 1140, 1152, 30


def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


A = torch.randn(batch, M, LD, device=dev)
B = torch.randn(batch, Kd, LD, device=dev)
C = torch.empty(batch, M, LD, device=dev)
fl = 2.0 * batch * M * N * Kd
for name, sA, sB in (('distinct operands', M * LD, Kd * LD), ('shared operands (cache resident)', 0, 0)):
    for tB, Bm in ((False, B), (True, B)):
        ms = bench(lambda: K.gemm(A, Bm, C, M, N, Kd, False, tB, LD, LD, LD, 1.0, 0.0, None, batch, sA, sB, M * LD))
        print('%-34s %s: %.3f ms  %.1f TFLOP/s' % (name, 'NT' if tB else 'NN', ms, fl / ms / 1e9))
