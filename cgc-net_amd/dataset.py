"""On-disk graph datasets of the reference: the per-epoch ``.pt`` ``Data`` files written by
``dataflow/prepare_cv_dataset.py:96-107,148`` and consumed by ``NucleiDatasetBatchOutput.get`` (``dataflow/data.py:330-354``).

Layout under ``root`` (``setting.root``):
  proto/fix_fuse_cia_knn/<epoch>/<fold>/<image>.pt   pre-sampled graphs, one directory per epoch (``dynamic_graph=False``)
  proto/cross_val/<fold>/<image>.pt                  full nucleus sets (``dynamic_graph=True``: re-sampled on every access)
Each file is a pickled ``torch_geometric.data.Data`` (torch-geometric 1.2.1: a plain object whose ``__dict__`` holds ``x``
[n, 16+2] = appearance features + the two coordinates, ``pos`` [n, 2], ``y`` [1], and None placeholders).  PyG is not
needed to read them: the unpickler below maps the class onto ``cgc_net_amd.data.Data``.

``get(idx)`` mirrors the reference's item: feature slice by ``feature_type`` ('ca' all columns, 'c' the coordinates, 'a' the
appearance features), optional re-sampling, ``patch_idx``, and -- host mode -- the k-NN graph (``radius_graph(pos, 100, None,
True, 8)``) and the z-scoring ``(x - mean) / std``.  With ``device_front_end=True`` the item stays raw (no edges, no
z-scoring): ``Batch.from_data_list(items, device=..., knn=dataset.knn, mean=dataset.mean, std=dataset.std)`` (F1/F2: one packed
copy, collate + k-NN on the GPU) finishes the whole batch at once; ``front_end_kwargs()`` returns exactly those arguments.
"""
import os
import os.path as osp
import pickle

import numpy as np
import torch

from .data import Data, radius_graph

CROSS_VAL = {1: {'train': ['fold_1', 'fold_2'], 'valid': ['fold_3']},      # dataflow/data.py:15-19
             2: {'train': ['fold_1', 'fold_3'], 'valid': ['fold_2']},
             3: {'train': ['fold_2', 'fold_3'], 'valid': ['fold_1']}}


class _PygUnpickler(pickle.Unpickler):
    """``torch_geometric.data[.data].Data`` / ``Batch`` -> our attribute bags (same ``__dict__`` protocol)."""

    def find_class(self, module, name):
        if module.startswith('torch_geometric.data') and name in ('Data', 'Batch'):
            from . import data as _d
            return getattr(_d, name)
        return super().find_class(module, name)


class _PygPickle(object):
    """``pickle_module`` for ``torch.load``: the stock pickle with the class mapping above."""
    __name__ = 'cgc_pyg_pickle'
    Unpickler = _PygUnpickler
    load = staticmethod(lambda f, **kw: _PygUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    DEFAULT_PROTOCOL = pickle.DEFAULT_PROTOCOL
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def load_pt(path):
    """Read one of the reference's ``.pt`` graph files (torch.save of a PyG ``Data``) without torch_geometric."""
    obj = torch.load(path, map_location='cpu', pickle_module=_PygPickle, weights_only=False)
    if isinstance(obj, dict):                                   # newer PyG versions pickle the mapping itself
        obj = Data(**obj)
    return obj


def save_pt(data, path):
    """Write a graph the way prepare_cv_dataset.py:105-107 does (``torch.save(data, <epoch>/<fold>/<image>.pt)``)."""
    d = osp.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(data, path)


class NucleiDatasetBatchOutput(torch.utils.data.Dataset):
    """dataflow/data.py:308-354 (+ the constructor of NucleiDataset, :112-163) on this package's ``Data``.

    ``mean`` / ``std``: the z-scoring vectors over ALL stored columns (the reference hard-codes per-fold tables,
    dataflow/data.py:21-48; ``feature_statistics`` computes them from the files); sliced by ``feature_type`` as there."""

    def __init__(self, root, feature_type='ca', split='train', sampling_ratio=0.5, dynamic_graph=False,
                 sampling_method='fuse', neighbour=8, max_edge_distance=100, crossval=1, mean=None, std=None,
                 fix_dir='fix_fuse_cia_knn', device_front_end=False, task='colon'):
        assert feature_type in ('ca', 'c', 'a') and split in ('train', 'valid')
        self.root, self.feature_type, self.split = root, feature_type, split
        self.sampling_ratio, self.dynamic_graph, self.sample_method = sampling_ratio, dynamic_graph, sampling_method
        self.max_neighbours, self.max_edge_distance, self.cross_val = neighbour, max_edge_distance, crossval
        self.device_front_end = device_front_end
        self.task = task                 # setting.name (dataflow/data.py:121): graphs under 100 nodes keep all nodes unless 'colon'
        self.epoch = self.val_epoch = 0
        self.processed_root = osp.join(root, 'proto', 'cross_val')
        self.processed_fix_data_root = osp.join(root, 'proto', fix_dir)
        folds = CROSS_VAL[crossval][split]
        listing_root = self.processed_root if (dynamic_graph or not osp.isdir(self.processed_fix_data_root)) \
            else osp.join(self.processed_fix_data_root, '0')
        self.idxlist = []
        for fold in folds:                                       # dataflow/data.py:159-161
            d = osp.join(listing_root, fold)
            if osp.isdir(d):
                # (sorted: the reference takes os.listdir order, which is file-system dependent; patch_idx values follow this list)
                self.idxlist.extend(osp.join(fold, f) for f in sorted(os.listdir(d)) if f.endswith('.pt'))
        self.mean = None if mean is None else self._slice_cols(torch.as_tensor(mean, dtype=torch.float32))
        self.std = None if std is None else self._slice_cols(torch.as_tensor(std, dtype=torch.float32))

    # -- protocol the training loop uses (train.py:36,173)
    def set_epoch(self, epoch):
        self.epoch = epoch

    def set_val_epoch(self, epoch):
        self.val_epoch = epoch

    def __len__(self):
        return len(self.idxlist)

    @property
    def knn(self):
        return (float(self.max_edge_distance), int(self.max_neighbours))

    def front_end_kwargs(self):
        """Arguments for ``Batch.from_data_list(items, device=..., **kw)`` / ``DataParallel(front_end=kw)``."""
        kw = dict(knn=self.knn)
        if self.mean is not None:
            kw.update(mean=self.mean, std=self.std)
        return kw

    def _slice_cols(self, t):
        if self.feature_type == 'c':
            return t[..., -2:]
        if self.feature_type == 'a':
            return t[..., :-2]
        return t

    def path_of(self, idx):
        epoch = self.epoch if self.split == 'train' else self.val_epoch
        if self.dynamic_graph:
            return osp.join(self.processed_root, self.idxlist[idx])
        return osp.join(self.processed_fix_data_root, str(epoch), self.idxlist[idx])

    def __getitem__(self, idx):
        data = load_pt(self.path_of(idx))
        data.x = self._slice_cols(data.x.to(torch.float32))
        if self.dynamic_graph:                                   # dataflow/data.py:338-344 (host-side, one image): the reference draws
            n = data.num_nodes                                   # and permutes also at ratio 1
            choice = self._sample(data.pos, n)
            for key, item in list(data):
                if torch.is_tensor(item) and item.dim() > 0 and item.size(0) == n:
                    data[key] = item[choice]
        data.patch_idx = torch.tensor([idx])
        if self.device_front_end:
            data.edge_index = None                               # built on the GPU for the whole batch
            return data
        data.edge_index = radius_graph(data.pos, self.max_edge_distance, None, True, self.max_neighbours)
        if self.mean is not None:
            data.x = (data.x - self.mean) / self.std             # dataflow/data.py:353
        return data

    def _sample(self, pos, n):
        """Reference-compatible draw on the host for ONE image (the table's int16 distances re-derived from ``pos``)."""
        return _sample_one_host(pos, n, self.sampling_ratio, self.sample_method, self.task)


def _sample_one_host(pos, n, ratio, method, task='colon'):
    """FarthestSampler / 'fuse' / 'random' on the host with the table's arithmetic (dataflow/data.py:195-223;
    common/utils.py:187-203; dataflow/construct_feature_graph.py:17-24), row by row instead of from a stored n x n file."""
    import random as pyrandom
    k = int(n * ratio)
    if task != 'colon' and n < 100:                              # dataflow/data.py:199-201
        k = n
    p = pos.detach().cpu().numpy().astype(np.float32)[:, :2]

    def farthest(kf):
        picks = np.zeros(kf, dtype=np.int64)
        if kf == 0:
            return picks
        picks[0] = np.random.randint(n)

        def row(i):
            dx, dy = p[:, 0] - p[i, 0], p[:, 1] - p[i, 1]
            return np.sqrt(dx ** 2 + dy ** 2).astype(np.int16)
        dist = row(picks[0])
        for i in range(1, kf):
            picks[i] = np.argmax(dist)
            dist = np.minimum(dist, row(picks[i]))
        return picks
    if method == 'farthest':
        return torch.from_numpy(farthest(k))
    if method == 'fuse':
        far = farthest(int(0.7 * k))
        taken = set(far.tolist())
        remain = [i for i in range(n) if i not in taken]
        rand = np.asarray(pyrandom.sample(remain, k - len(far)), dtype=np.int64)
        return torch.from_numpy(np.concatenate((far, rand), 0))
    return torch.from_numpy(np.random.choice(n, k, replace=False))


def feature_statistics(dataset, max_items=None):
    """Column mean / std over the stored graphs (what the reference's hard-coded _MEAN_CIA/_STD_CIA tables hold)."""
    s = s2 = None
    cnt = 0
    for i in range(len(dataset) if max_items is None else min(max_items, len(dataset))):
        x = load_pt(dataset.path_of(i)).x.double()
        s = x.sum(0) if s is None else s + x.sum(0)
        s2 = (x * x).sum(0) if s2 is None else s2 + (x * x).sum(0)
        cnt += x.shape[0]
    mean = s / cnt
    std = (s2 / cnt - mean * mean).clamp_min(0).sqrt()
    return mean.float(), std.float()
