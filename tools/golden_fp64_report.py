#!/usr/bin/env python
"""Per-parameter gradient error of the HIP path against the REFERENCE-generated float64 fixtures (tests/golden/*_fp64.npz), all eight
cases, no assertion on the gradient bar: prints the worst parameters.  GPU only.  CGC_LIB selects a variant library.
``--big-route``: every product forced onto the 128 x 128 pipelined route (cgc_gemm_tuning(11)); ``--split``: CGC_GEMM_SPLIT_BF16 (the
products on that route as six bf16 MFMA pairs); ``--half``: mode CGC_GEMM_SPLIT_F16 (three fp16 pairs of scaled operands) -- the number of products that took the split kernel is printed per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
os.environ['CGC_GEMM_SPLIT_BF16'] = '2' if '--half' in sys.argv else '1' if '--split' in sys.argv else '0'      # explicit (the module's default is not exact since round 6)
import discrete  # noqa: E402
from util import CASES  # noqa: E402
from cgc_net_amd import kernels  # noqa: E402

os.environ['CGC_PARITY_REPORT'] = '1'
K = kernels.get()
if '--big-route' in sys.argv:
    K.lib.cgc_gemm_tuning(11)
print('# routing: %s; GEMM mode: %s' % ('every product on the 128 x 128 route' if '--big-route' in sys.argv else 'automatic',
                                        'three fp16 pairs' if '--half' in sys.argv else 'split bf16' if '--split' in sys.argv else 'exact fp32'))
for name in CASES:
    before = int(K.lib.cgc_gemm_split_count()) + int(K.lib.cgc_gemm_half_count())
    try:
        discrete.compare_with_reference_fp64(name, tol_grad=1e9)
    except AssertionError as e:
        print(name, 'FAILED', str(e)[:300])
    print('%s: products on the 16-bit kernels: %d' % (name, int(K.lib.cgc_gemm_split_count()) + int(K.lib.cgc_gemm_half_count()) - before))
