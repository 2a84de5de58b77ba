#!/usr/bin/env python
"""Golden case for the GEXF export of the assignment matrices  -- BUILD CONTAINER ONLY.

Runs the REFERENCE's ``output_to_gexf`` (common/utils.py:48-79) on a small seeded case and stores inputs + the parsed
content of the file it wrote (node attributes, edges) in tests/golden/gexf_case.json.  common/utils.py is written against
networkx 2.x (``nx.from_numpy_matrix``); the image has networkx 3.4, where the same function is called ``from_numpy_array``:
the old NAME is aliased for the duration of this script (nothing else of networkx or of the reference is touched)."""
import json
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, 'pyg_standin'), ROOT]
sys.path.append(os.environ.get('CGC_REFERENCE', '/root/reference'))

import networkx as nx  # noqa: E402
import torch_geometric.nn as _pnn  # noqa: E402  (stand-in)
import torch_geometric.utils as _put  # noqa: E402
for _m, _n in ((_put, 'sparse_to_dense'), (_pnn, 'radius_graph')):      # imported by common/utils.py, unused here
    if not hasattr(_m, _n):
        setattr(_m, _n, None)
if not hasattr(nx, 'from_numpy_matrix'):
    nx.from_numpy_matrix = nx.from_numpy_array
from common import utils as refutils  # noqa: E402  (the reference)

rng = np.random.RandomState(3)
n, c1, c2 = 23, 6, 3
coord = rng.uniform(0, 500, size=(n, 2)).astype(np.float32)
adj = (rng.uniform(size=(n, n)) < 0.15).astype(np.float32)
adj = np.maximum(adj, adj.T)
np.fill_diagonal(adj, 1.0)
a1 = rng.uniform(size=(n, c1)).astype(np.float32)
a2 = rng.uniform(size=(c1, c2)).astype(np.float32)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, 'g.gexf')
    refutils.output_to_gexf(coord, adj, [a1, a2], path)
    G = nx.read_gexf(path)
nodes = {str(k): {a: (float(v) if isinstance(v, float) else int(v)) for a, v in attr.items() if a != 'label'} for k, attr in G.nodes(data=True)}
edges = sorted([sorted([int(u), int(v)]) + [float(d.get('weight', 1.0))] for u, v, d in G.edges(data=True)])
json.dump({'coord': coord.tolist(), 'adj': adj.tolist(), 'assign': [a1.tolist(), a2.tolist()], 'nodes': nodes, 'edges': edges},
          open(os.path.join(HERE, 'gexf_case.json'), 'w'))
print('nodes', len(nodes), 'edges', len(edges), 'attrs', sorted(next(iter(nodes.values()))))
