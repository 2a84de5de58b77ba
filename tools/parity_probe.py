#!/usr/bin/env python
"""Per-parameter gradient error of the HIP path and of the fp32 oracle against the fp64 oracle (tests/test_bench_size_parity_gpu.py's
yardstick), for a chosen batch size / flag set: python tools/parity_probe.py --batch 32 [--plain] [--seed 0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['CGC_PARITY_REPORT'] = '1'
import test_bench_size_parity_gpu as T  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


B, seed = arg('--batch', 32), arg('--seed', 0)
flags = dict() if '--plain' in sys.argv else dict(norm_adj=True, jk=True)
if '--jk-only' in sys.argv:
    flags = dict(jk=True)
if '--norm-only' in sys.argv:
    flags = dict(norm_adj=True)
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=seed)
b = Batch.from_data_list([ds[i] for i in range(B)])
try:
    T._compare_model(b, arg('--maxn', 11404), 16, flags)
except AssertionError as e:
    print('FAILED', str(e)[:600])
