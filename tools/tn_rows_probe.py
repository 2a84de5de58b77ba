"""Time the row-split weight-gradient contraction out[Fa,Fb] = A[n,Fa]^T B[n,Fb] (ops.gemm_tn_rows) for the shapes of a C3 step at
4 / 8 / 16 / 32 graphs per GPU, for the automatic number of row slices and for forced ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgc_net_amd import ops

dev = 'cuda:0'


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


for n in (7313, 14289, 29690, 58761):
    print('n =', n)
    for Fa, lda, Fb, ldb in ((1140, 1152, 40, 40), (1140, 1152, 1140, 1152), (20, 20, 1140, 1152), (1140, 1152, 20, 20), (20, 20, 20, 20)):
        A = torch.randn(n, lda, device=dev)
        B = torch.randn(n, ldb, device=dev)
        out = torch.empty(Fa, Fb, device=dev)
        auto = ops._split_parts(Fa, Fb, n)
        row = '  [%4d x %4d]  auto parts %3d: %7.1f us |' % (Fa, Fb, auto, bench(lambda: ops.gemm_tn_rows(A, lda, Fa, B, ldb, Fb, n, out)))
        for parts in (1, 4, 8, 14, 28, 56, 112, 224):
            if n // parts < 64:
                continue
            ops._split_parts.__defaults__[0][(Fa, Fb, n)] = parts
            row += ' %d: %.1f' % (parts, bench(lambda: ops.gemm_tn_rows(A, lda, Fa, B, ldb, Fb, n, out)))
        del ops._split_parts.__defaults__[0][(Fa, Fb, n)]
        print(row)
