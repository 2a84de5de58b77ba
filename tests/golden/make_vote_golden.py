#!/usr/bin/env python
"""Golden vectors for the image-level vote (common/metric.py:20-50): runs the REFERENCE's ImgLevelResult (imported from
/root/reference, its anonymised GROUND_TRUTH table replaced by synthetic image names) on seeded patch predictions and writes
inputs + expected (acc, binary_acc) to tests/golden/vote_cases.json.  Build container only; the fixture travels."""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True      # /root/reference is read-only: no __pycache__ may be left behind there
sys.path.insert(0, '/root/reference')
from common import metric  # noqa: E402

rng = np.random.RandomState(0)
cases = []
for case in range(6):
    n_img = int(rng.randint(3, 12))
    names = ['img%02d_grade_%d' % (i, int(rng.randint(1, 4))) for i in range(n_img)]
    metric.GROUND_TRUTH = {1: names, 2: names, 3: names}
    res = metric.ImgLevelResult(types.SimpleNamespace(cross_val=1))
    patches, labels = [], []
    for i, nm in enumerate(names):
        for p in range(int(rng.randint(1, 9))):
            patches.append('/data/proto/%s_patch_%d.pt' % (nm, p))
            labels.append(int(rng.randint(0, 3)))
    half = len(patches) // 2
    for nm, lb in zip(patches[:half], labels[:half]):
        res.patch_result(nm, lb)
    res.batch_patch_result(patches[half:], labels[half:])
    acc, bacc = res.final_result()
    cases.append({'ground_truth': names, 'patches': patches, 'labels': labels, 'acc': float(acc), 'binary_acc': float(bacc)})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'vote_cases.json')
json.dump(cases, open(out, 'w'), indent=0)
print('wrote', out, len(cases), 'cases')
