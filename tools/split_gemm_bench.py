#!/usr/bin/env python
"""The six dominant products of a 32-graph step (profiles/r04_gemm_calls_by_shape.txt), stand-alone, in both modes of cgc_gemm_f32_ws:
exact (fp32 MFMA chain), split (six bf16 MFMA pairs, csrc/gemm_split.hip) and half (three fp16 pairs of scaled operands, csrc/gemm_half.hip;
its time includes the operand-maximum pass).  Operands on the row strides the step uses.  Per product:
time of a launch (mean over `reps` back-to-back launches after a long warm-up: the clock ramps over milliseconds), fp32-equivalent
TFLOP/s, fraction of the fp32 MFMA peak (157.3) and -- split / half mode -- of the 16-bit pipe: (6 | 3) x 2MNK / t / 2500 TF.
usage: python tools/split_gemm_bench.py [reps] [graphs]     CGC_LIB selects a variant library."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels  # noqa: E402

dev = 'cuda:0'
K = kernels.get()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
C, LD = 1140, 1152
rng = np.random.RandomState(0)
counts = rng.randint(1440, 2161, size=B)
n, nmax = int(counts.sum()), int(counts.max())
gptr = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
torch.manual_seed(0)
X = torch.randn(n, LD, device=dev)            # an [n, 1140] activation on 1152-float rows
Y = torch.randn(n, LD, device=dev)
W = torch.randn(C, LD, device=dev) * 0.05
sq = torch.randn(B, C, C, device=dev) * 0.05    # a [1140, 1140] matrix per graph
xe = torch.randn(n, 40, device=dev)
we = torch.randn(40, LD, device=dev) * 0.05
xs = torch.randn(n, 20, device=dev)
ws20 = torch.randn(B, C, 20, device=dev) * 0.05
out_n = torch.empty(n, LD, device=dev)
out_sq = torch.empty(B, C, C, device=dev)
chunk = (-(-n // 6) + 31) // 32 * 32           # six row chunks, as the step cuts its weight gradient
parts = -(-n // chunk)
out_parts = torch.empty(parts, C, C, device=dev)
cases_all = [
    ('Linear fwd   NN flat + extra K 40', lambda: K.gemm(X, W, out_n, n, C, C, False, False, LD, LD, LD, 1.0, 0.0, None, extra=[(xe, we, 40, LD, 40, 0, 0)]), n),
    ('Linear dx    NN flat', lambda: K.gemm(Y, W, out_n, n, C, C, False, False, LD, LD, LD), n),
    ('dP = S dA\'   NN ragged M', lambda: K.gemm(X, sq, out_n, 0, C, C, False, False, LD, C, LD, 1.0, 0.0, None, B, 0, C * C, 0, gptr, 1, nmax, n), n),
    ('dS += P dA\'^T + X dX\'^T  NT ragged M, extra K 20, beta 1',
     lambda: K.gemm(Y, sq, out_n, 0, C, C, False, True, LD, C, LD, 1.0, 1.0, None, B, 0, C * C, 0, gptr, 1, nmax, n, extra=[(xs, ws20, 20, 20, 20, 0, C * 20)]), n),
    ('dW = X^T dY  TN uniform chunks', lambda: K.gemm(X, Y, out_parts, C, C, n, True, False, LD, LD, C, 1.0, 0.0, None, parts, 0, 0, C * C, None, 3, chunk, n), n),
    ('S^T P        TN ragged K', lambda: K.gemm(X, Y, out_sq, C, C, 0, True, False, LD, LD, C, 1.0, 0.0, None, B, 0, 0, C * C, gptr, 2, nmax, n), n),
]
sel = os.environ.get('SPLIT_BENCH_CASES')          # e.g. 1,4: only these products
cases = [c for i, c in enumerate(cases_all) if sel is None or str(i) in sel.split(',')]
MODES = [int(m) for m in os.environ.get('SPLIT_BENCH_MODES', '0,1,2').split(',')]     # 0 exact, 1 six bf16 pairs, 2 three fp16 pairs
NAMES = {0: 'exact', 1: 'split', 2: 'half'}
PAIRS = {1: 6, 2: 3}
tot = {m: 0.0 for m in MODES}
print('%d graphs, %d rows; lib=%s' % (B, n, os.path.basename(kernels.lib_path())))
for name, fn, rows in cases:
    fl = 2.0 * rows * C * C
    line = '%-62s' % name
    for mode in MODES:
        K.gemm_mode = mode
        out_n.zero_()
        for _ in range(25):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        tot[mode] += ms
        tf = fl / ms / 1e9
        line += '  %s %7.1f us %6.1f TF (%.3f of fp32 MFMA%s)' % (NAMES[mode], ms * 1e3, tf, tf / 157.3,
                                                                   '' if mode == 0 else '; 16-bit pipe %.3f' % (PAIRS[mode] * tf / 2500.0))
    K.gemm_mode = 0
    print(line)
print('six products: ' + ', '.join('%s %.1f us' % (NAMES[m], tot[m] * 1e3) for m in MODES))
