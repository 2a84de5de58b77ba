#!/usr/bin/env python
"""Per-parameter gradient error of the HIP path against the REFERENCE-generated float64 fixtures (tests/golden/*_fp64.npz), all five
cases, no assertion on the gradient bar: prints the worst parameters.  GPU only.  CGC_LIB selects a variant library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import discrete  # noqa: E402
from util import CASES  # noqa: E402

os.environ['CGC_PARITY_REPORT'] = '1'
for name in CASES:
    try:
        discrete.compare_with_reference_fp64(name, tol_grad=1e9)
    except AssertionError as e:
        print(name, 'FAILED', str(e)[:300])
