#!/usr/bin/env python
"""Per-parameter gradient error of the HIP path against the REFERENCE-generated float64 fixtures (tests/golden/*_fp64.npz), all eight
cases, no assertion on the gradient bar: prints the worst parameters.  GPU only.  CGC_LIB selects a variant library.
``--big-route``: every product forced onto the 128 x 128 pipelined route (cgc_gemm_tuning(11)); ``--split``: CGC_GEMM_SPLIT_BF16 (the
products on that route as six bf16 MFMA pairs) -- the number of products that took the split kernel is printed per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
os.environ['CGC_GEMM_SPLIT_BF16'] = '1' if '--split' in sys.argv else '0'      # explicit either way (the module's default is split since round 6)
import discrete  # noqa: E402
from util import CASES  # noqa: E402
from cgc_net_amd import kernels  # noqa: E402

os.environ['CGC_PARITY_REPORT'] = '1'
K = kernels.get()
if '--big-route' in sys.argv:
    K.lib.cgc_gemm_tuning(11)
print('# routing: %s; GEMM mode: %s' % ('every product on the 128 x 128 route' if '--big-route' in sys.argv else 'automatic',
                                        'split bf16' if '--split' in sys.argv else 'exact fp32'))
for name in CASES:
    before = int(K.lib.cgc_gemm_split_count())
    try:
        discrete.compare_with_reference_fp64(name, tol_grad=1e9)
    except AssertionError as e:
        print(name, 'FAILED', str(e)[:300])
    print('%s: products on the split kernel: %d' % (name, int(K.lib.cgc_gemm_split_count()) - before))
