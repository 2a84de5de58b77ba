#!/usr/bin/env python
"""Golden vectors for the node samplers  -- BUILD CONTAINER ONLY.

Imports the REFERENCE's ``FarthestSampler`` and ``filter_sampled_indice`` (common/utils.py:187-203; under the
torch_geometric stand-in, because common/utils.py imports PyG names it does not need here) and runs them on the int16
distance table the reference stores next to every coordinate file.  ``euc_dist`` itself
(dataflow/construct_feature_graph.py:17-24) lives in a module that imports skimage/cv2 (absent), so its three arithmetic lines
are repeated here verbatim in meaning: float32 coordinates, ``sqrt((x.T-x)**2 + (y.T-y)**2).astype(np.int16)``.
The 'fuse' composition follows dataflow/data.py:210-217 (``far_num = int(0.7*k)``, ``random.sample(remain, rand_num)``,
``np.concatenate((far, rand))``).  Output: tests/golden/sampler_cases.npz (coordinates, k, seeds, expected indices)."""
import os
import random
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, 'pyg_standin'), ROOT]
REF = os.environ.get('CGC_REFERENCE', '/root/reference')
sys.path.append(REF)

import torch_geometric.nn as _pnn  # noqa: E402  (stand-in)
import torch_geometric.utils as _put  # noqa: E402
for _m, _n in ((_put, 'sparse_to_dense'), (_pnn, 'radius_graph')):      # imported by common/utils.py, unused by the samplers
    if not hasattr(_m, _n):
        setattr(_m, _n, None)
from common import utils as refutils  # noqa: E402  (the reference)


def table_of(coords):
    """dataflow/construct_feature_graph.py:17-24 on the float32 coordinate array (:122 stores float32)."""
    arr = coords.astype(np.float32)
    arr_x = (arr[:, 0, np.newaxis].T - arr[:, 0, np.newaxis]) ** 2
    arr_y = (arr[:, 1, np.newaxis].T - arr[:, 1, np.newaxis]) ** 2
    return np.sqrt(arr_x + arr_y).astype(np.int16)


def layouts():
    rng = np.random.RandomState(0)
    yield 'uniform_500', rng.uniform(0, 3584, size=(500, 2)).astype(np.float32)
    yield 'uniform_2300', rng.uniform(0, 3584, size=(2300, 2)).astype(np.float32)
    gx, gy = np.meshgrid(np.arange(20) * 37.0, np.arange(20) * 37.0)                 # lattice: every distance ties many times
    yield 'lattice_400', np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    c = rng.randint(0, 60, size=(300, 2)).astype(np.float32)                          # integer coordinates: perfect squares, duplicates
    yield 'integer_300', c
    yield 'clustered_700', (rng.standard_normal((700, 2)) * 40 + rng.randint(0, 5, size=(700, 1)) * 600).astype(np.float32)


out = {}
names = []
sampler = refutils.FarthestSampler()
for name, coords in layouts():
    n = coords.shape[0]
    table = table_of(coords)
    k = int(n * 0.5)
    for seed in (1, 2):
        np.random.seed(seed)
        far = sampler(table, k)                                  # 'farthest': dataflow/data.py:205-208
        np.random.seed(seed)
        random.seed(seed)
        far_num = int(0.7 * k)
        rand_num = k - far_num
        far_indice = sampler(table, far_num)                      # 'fuse': dataflow/data.py:210-217
        remain_item = refutils.filter_sampled_indice(far_indice, n)
        rand_indice = np.asarray(random.sample(remain_item, rand_num))
        fuse = np.concatenate((far_indice, rand_indice), 0)
        key = '%s/s%d' % (name, seed)
        out[key + '/farthest'] = far.astype(np.int32)
        out[key + '/fuse'] = fuse.astype(np.int32)
    out[name + '/coords'] = coords
    names.append(name)
out['names'] = np.asarray(names)
np.savez_compressed(os.path.join(HERE, 'sampler_cases.npz'), **out)
print('wrote sampler_cases.npz:', names)
