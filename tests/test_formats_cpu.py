"""F4: on-disk formats of the reference -- the per-epoch ``.pt`` graph files (pickled torch_geometric ``Data``),
the dataset class that walks them, and the GEXF export of the assignment matrices (pinned by a fixture produced by the
reference's own ``output_to_gexf``, tests/golden/make_gexf_golden.py)."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import dataset as dsmod
from cgc_net_amd.data import Batch, Data, DataListLoader, radius_graph
from cgc_net_amd.evalio import cluster_labels, output_to_gexf
from util import GOLDEN


def _write_pyg_pickles(root, epochs=2, per_fold=3, feat=18, seed=0):
    """Files exactly as torch-geometric 1.2.1 pickles them: class ``torch_geometric.data.data.Data``, state = its __dict__
    (x, edge_index, edge_attr, y, pos with None placeholders).  A throw-away module of that name provides the class while
    saving; it is removed again, so loading happens WITHOUT any torch_geometric module."""
    mod_names = ['torch_geometric', 'torch_geometric.data', 'torch_geometric.data.data']
    saved = {m: sys.modules.get(m) for m in mod_names}
    mods = [types.ModuleType(m) for m in mod_names]
    for m in mods:
        sys.modules[m.__name__] = m

    class PygData(object):
        def __init__(self, **kw):
            self.x = self.edge_index = self.edge_attr = self.y = self.pos = None
            self.__dict__.update(kw)
    PygData.__name__ = PygData.__qualname__ = 'Data'
    PygData.__module__ = 'torch_geometric.data.data'
    mods[2].Data = PygData
    rng = np.random.RandomState(seed)
    truth = {}
    try:
        for fold in ('fold_1', 'fold_2', 'fold_3'):
            for i in range(per_fold):
                n_full = int(rng.randint(60, 90))
                full = rng.uniform(0, 900, size=(n_full, 2)).astype(np.float32)
                feats = np.concatenate([rng.standard_normal((n_full, feat - 2)).astype(np.float32) * 3 + 1, full], 1)
                y = int(rng.randint(3))
                name = 'img%d_grade_%d.pt' % (i, y + 1)
                os.makedirs(os.path.join(root, 'proto', 'cross_val', fold), exist_ok=True)
                torch.save(PygData(x=torch.from_numpy(feats), pos=torch.from_numpy(full), y=torch.tensor([y])),
                           os.path.join(root, 'proto', 'cross_val', fold, name))
                for ep in range(epochs):
                    keep = np.sort(rng.choice(n_full, n_full // 2, replace=False))
                    sub = PygData(x=torch.from_numpy(feats[keep]), pos=torch.from_numpy(full[keep]), y=torch.tensor([y]))
                    sub.edge_index = radius_graph(sub.pos, 100, None, True, 8)      # prepare_cv_dataset.py:102
                    d = os.path.join(root, 'proto', 'fix_fuse_cia_knn', str(ep), fold)
                    os.makedirs(d, exist_ok=True)
                    torch.save(sub, os.path.join(d, name))
                    truth[(ep, fold, name)] = (feats[keep], full[keep], y)
    finally:
        for m in mod_names:
            if saved[m] is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = saved[m]
    assert 'torch_geometric.data.data' not in sys.modules or saved['torch_geometric.data.data'] is not None
    return truth


def test_reads_reference_pt_files_and_epoch_layout(tmp_path):
    truth = _write_pyg_pickles(str(tmp_path))
    mean, std = np.arange(18, dtype=np.float32), np.arange(18, dtype=np.float32) + 1.0
    for split, folds in (('train', ['fold_1', 'fold_2']), ('valid', ['fold_3'])):
        ds = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split=split, crossval=1, mean=mean, std=std)
        assert len(ds) == 3 * len(folds) and all(p.split('/')[0] in folds for p in ds.idxlist)
        for ep in (0, 1):
            ds.set_epoch(ep), ds.set_val_epoch(ep)
            for i in range(len(ds)):
                d = ds[i]
                fold, name = ds.idxlist[i].split('/')
                feats, pos, y = truth[(ep, fold, name)]
                assert isinstance(d, Data) and int(d.y) == y and int(d.patch_idx) == i
                assert torch.equal(d.pos, torch.from_numpy(pos))
                assert torch.allclose(d.x, (torch.from_numpy(feats) - torch.from_numpy(mean)) / torch.from_numpy(std))
                assert torch.equal(d.edge_index, radius_graph(d.pos, 100, None, True, 8))
    # feature_type slices (dataflow/data.py:336-339), loader protocol, collate
    ds_c = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'c', split='valid', mean=mean, std=std)
    ds_a = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'a', split='valid', mean=mean, std=std)
    assert ds_c[0].x.shape[1] == 2 and ds_a[0].x.shape[1] == 16
    items = next(iter(DataListLoader(ds_a, batch_size=3)))
    b = Batch.from_data_list(items)
    assert b.x.shape[1] == 16 and b.num_graphs == 3 and b.batch.numel() == b.x.shape[0]
    m, s = dsmod.feature_statistics(ds_a)
    assert m.shape == (18,) and bool((s > 0).all())


def test_dynamic_graph_resamples_with_the_reference_sampler(tmp_path):
    import random
    _write_pyg_pickles(str(tmp_path))
    ds = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', dynamic_graph=True, sampling_method='fuse')
    full = dsmod.load_pt(ds.path_of(0))
    np.random.seed(5), random.seed(5)
    d = ds[0]
    n = full.x.shape[0]
    assert d.x.shape[0] == int(n * 0.5) and d.edge_index.max() < d.x.shape[0]
    np.random.seed(5), random.seed(5)
    choice = dsmod._sample_one_host(full.pos, n, 0.5, 'fuse')
    assert torch.equal(d.pos, full.pos[choice]) and len(set(choice.tolist())) == choice.numel()


def test_dynamic_graph_sampling_rules_of_the_reference(tmp_path):
    """dataflow/data.py:195-223: graphs under 100 nodes keep every node unless the task is 'colon'; with dynamic_graph the draw
    (and the permutation it implies) happens at ratio 1 as well."""
    import random
    _write_pyg_pickles(str(tmp_path))                        # graphs of 60..89 nodes
    full = dsmod.load_pt(dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', dynamic_graph=True).path_of(0))
    n = full.x.shape[0]
    other = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', dynamic_graph=True, sampling_method='random', task='prostate')
    assert other[0].x.shape[0] == n                          # < 100 nodes, not 'colon': all of them
    colon = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', dynamic_graph=True, sampling_method='random')
    assert colon[0].x.shape[0] == int(n * 0.5)
    one = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', dynamic_graph=True, sampling_method='random', sampling_ratio=1.0)
    np.random.seed(3), random.seed(3)
    d = one[0]
    np.random.seed(3), random.seed(3)
    choice = dsmod._sample_one_host(full.pos, n, 1.0, 'random')
    assert sorted(choice.tolist()) == list(range(n)) and torch.equal(d.pos, full.pos[choice]) and not torch.equal(d.pos, full.pos)


def test_device_front_end_items_stay_raw(tmp_path, torch_kernels):
    _write_pyg_pickles(str(tmp_path))
    mean, std = np.zeros(18, dtype=np.float32) + 2.0, np.ones(18, dtype=np.float32) * 4.0
    host = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', mean=mean, std=std)
    raw = dsmod.NucleiDatasetBatchOutput(str(tmp_path), 'ca', split='valid', mean=mean, std=std, device_front_end=True)
    want = Batch.from_data_list([host[i] for i in range(3)])
    got = Batch.from_data_list([raw[i] for i in range(3)], device='cpu', **raw.front_end_kwargs())     # F1/F2 path (CPU seam)
    assert torch.allclose(got.x, want.x) and torch.equal(got.batch, want.batch)
    assert torch.equal(got.edge_index, want.edge_index)


def test_gexf_export_matches_reference_fixture(tmp_path):
    import networkx as nx
    case = json.load(open(os.path.join(GOLDEN, 'gexf_case.json')))
    coord, adj = np.asarray(case['coord'], dtype=np.float32), np.asarray(case['adj'], dtype=np.float32)
    assign = [np.asarray(a, dtype=np.float32) for a in case['assign']]

    def parsed(path):
        G = nx.read_gexf(path)
        nodes = {str(k): {a: v for a, v in attr.items() if a != 'label'} for k, attr in G.nodes(data=True)}
        edges = sorted([sorted([int(u), int(v)]) + [float(d.get('weight', 1.0))] for u, v, d in G.edges(data=True)])
        return nodes, edges
    p1 = str(tmp_path / 'dense.gexf')
    output_to_gexf(coord, adj, assign, p1)
    nodes, edges = parsed(p1)
    assert edges == [list(e) for e in case['edges']]
    assert set(nodes) == set(case['nodes'])
    for k, attr in case['nodes'].items():
        assert nodes[k]['assign_1'] == attr['assign_1'] and nodes[k]['assign_2'] == attr['assign_2']
        assert abs(nodes[k]['x'] - attr['x']) < 1e-4 and abs(nodes[k]['y'] - attr['y']) < 1e-4
    # the sparse form (edge_index instead of an n x n matrix) describes the same graph
    r, c = np.nonzero(adj)
    p2 = str(tmp_path / 'sparse.gexf')
    output_to_gexf(torch.from_numpy(coord), torch.from_numpy(np.stack([r, c])), [torch.from_numpy(a) for a in assign], p2)
    assert parsed(p2) == (nodes, edges)
    lab = cluster_labels(assign)
    assert lab['assign_2'].tolist() == [int(np.argmax(assign[1][j])) for j in np.argmax(assign[0], 1)]
