"""Host-side graph containers and the seeded synthetic cell-graph generator.

Mirrors the slice of the torch_geometric data API that the reference's hot path
and training loop touch:

* ``Data(x, pos, y, edge_index, patch_idx)``        -- dataflow/data.py:330-354 builds these
* ``Batch.from_data_list`` (node-offset concat + sorted ``batch`` vector)
                                                     -- consumed at model/network.py:239-240
* ``DataListLoader`` (a DataLoader whose collate is the identity: yields python lists of Data)
                                                     -- train.py:52,175 iterate it
* ``radius_graph(pos, r, batch, loop, max_num_neighbors)``
                                                     -- dataflow/data.py:348, prepare_cv_dataset.py:102

The host paths mirror torch_geometric; ``Batch.from_data_list(..., device=)`` and ``radius_graph`` on CUDA positions are the
device front-end (csrc/collate.hip, csrc/knn.hip).  The CSR the model consumes is built in graph.py.
"""
import numpy as np
import torch
from scipy.spatial import cKDTree


class Data(object):
    """Attribute bag with the torch_geometric ``Data`` surface the reference uses."""

    def __init__(self, x=None, edge_index=None, y=None, pos=None, **kwargs):
        self.x, self.edge_index, self.y, self.pos = x, edge_index, y, pos
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        """The data attributes, as torch_geometric lists them.  Underscore attributes (``_gptr``, ``_dense_rows``, ``_spatial`` ...) are
        host-side notes of this package about the object -- they travel with ``to()`` / ``reorder_nodes`` but are not data."""
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith('_')]

    def __iter__(self):  # dataflow/data.py:344 iterates ``for key, item in data``
        for k in self.keys:
            yield k, getattr(self, k)

    def _all_items(self):
        return [(k, v) for k, v in self.__dict__.items() if v is not None]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __contains__(self, k):
        return getattr(self, k, None) is not None

    @property
    def num_nodes(self):
        return self.x.shape[0] if self.x is not None else self.pos.shape[0]

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    def to(self, device, non_blocking=False):
        out = self.__class__()
        for k, v in self._all_items():
            out[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out

    def __repr__(self):
        parts = ['%s=%s' % (k, list(v.shape) if torch.is_tensor(v) else v) for k, v in self]
        return '%s(%s)' % (self.__class__.__name__, ', '.join(parts))


class Batch(Data):
    """Several graphs as one disconnected graph; ``batch[i]`` = graph id of node i (sorted)."""

    @staticmethod
    def from_data_list(data_list, device=None, knn=None, mean=None, std=None, spatial=False):
        """Collate a python list of ``Data`` (what ``DataListLoader`` yields, train.py:52,175).

        Plain call (``device`` None): the torch_geometric behaviour, on the host.  With ``device``: the loader front-end on
        the accelerator -- every per-graph array is packed into ONE pinned staging buffer, copied with ONE host-to-device
        transfer, and finished by one kernel (``batch`` vector, node offsets on ``edge_index``, and -- if ``mean``/``std``
        are given -- the z-scoring ``x = (x - mean) / std`` of dataflow/data.py:353).  ``knn = (radius, max_neighbours)``
        builds ``edge_index`` on the device from ``pos`` instead (dataflow/data.py:348 ``radius_graph(pos, r, None, True,
        k)`` per graph), so the items need not carry edges at all.  ``spatial=True`` (with ``knn``) first lists the nodes of
        every graph grid cell by grid cell (``spatial_order``): the model is invariant to the node order, the wide neighbour
        aggregation is up to 35 % faster on graphs of thousands of nodes when neighbours are also neighbours in memory."""
        if device is not None:
            return _collate_on_device(data_list, torch.device(device), knn, mean, std, spatial)
        assert knn is None and mean is None and std is None and not spatial, \
            'host collate: items arrive finished (dataflow/data.py:330-354)'
        out = Batch()
        keys = [k for k in data_list[0].keys if not k.startswith('_')]      # (underscore attributes are host-side notes, not data)
        offset, cat, batch_vec = 0, {k: [] for k in keys}, []
        for g, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = d[k]
                if k == 'edge_index':
                    v = v + offset          # cumulative node offset
                cat[k].append(v)
            batch_vec.append(torch.full((n,), g, dtype=torch.long))
            offset += n
        for k in keys:
            if torch.is_tensor(cat[k][0]):
                out[k] = torch.cat(cat[k], dim=1 if k == 'edge_index' else 0)
            else:
                out[k] = cat[k]
        out.batch = torch.cat(batch_vec)
        out.num_graphs = len(data_list)
        out._node_counts = [d.num_nodes for d in data_list]   # host-side: lets graph.py skip a device sync
        # per-graph row offsets: travel to the device with the batch (.to()), so that the model does not start every step with a
        # blocking host-to-device copy of its own
        out._gptr = torch.tensor([0] + list(np.cumsum(out._node_counts)), dtype=torch.int32)
        # the edges of graph g are edge_index[:, _eptr[g]:_eptr[g+1]] (this collate concatenates per graph): lets the CSR build work
        # graph by graph in two launches (graph.BatchGraph -> cgc_graph_build_local); _etotal guards against an edge_index that
        # was replaced afterwards
        if 'edge_index' in keys and all(torch.is_tensor(d.edge_index) for d in data_list):
            ec = [int(d.edge_index.shape[1]) for d in data_list]
            out._eptr = torch.tensor([0] + list(np.cumsum(ec)), dtype=torch.int32)
            out._etotal, out._emax = int(sum(ec)), int(max(ec))
        # every item says its nodes are listed grid cell by grid cell (spatial_order): the wide aggregation may stage neighbour unions
        # of consecutive rows in LDS (graph.BatchGraph.spatial -> cgc_spmm_graphs visit bit 2)
        if all(getattr(d, '_spatial', False) for d in data_list):
            out._spatial = True
        return out


class _Stager(object):
    """Two pinned staging buffers used alternately: the host fills one while the copy out of the other may still be in
    flight (an event per buffer guards the reuse)."""

    def __init__(self):
        self.buf, self.evt, self.turn = [None, None], [None, None], 0

    def get(self, nbytes, pinned):
        self.turn ^= 1
        t = self.turn
        if self.evt[t] is not None:
            self.evt[t].synchronize()
        if self.buf[t] is None or self.buf[t].numel() < nbytes or self.buf[t].is_pinned() != pinned:
            self.buf[t] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=pinned)
        return self.buf[t][:nbytes], t

    def sent(self, t, device):
        if device.type == 'cuda':
            self.evt[t] = torch.cuda.Event()
            self.evt[t].record(torch.cuda.current_stream(device))


_stager = _Stager()


def _collate_on_device(data_list, device, knn, mean, std, spatial=False):
    from . import kernels
    K = kernels.get()
    B = len(data_list)
    counts = [d.num_nodes for d in data_list]
    n = sum(counts)
    first = data_list[0]
    keys = [k for k in first.keys if not (k == 'edge_index' and knn is not None)]
    plan = []                                          # (key, per-graph tensors, cat dim)
    for k in keys:
        v = first[k]
        if torch.is_tensor(v):
            plan.append((k, [d[k] if k != 'x' else d[k].to(torch.float32) for d in data_list], 1 if k == 'edge_index' else 0))
    has_edges = any(k == 'edge_index' for k, _, _ in plan)
    gptr_h = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32)
    small = [('_gptr', gptr_h)]
    if has_edges:
        ecounts = [d.edge_index.shape[1] for d in data_list]
        small.append(('_eptr', torch.tensor(np.concatenate([[0], np.cumsum(ecounts)]), dtype=torch.int32)))
    for name, v in (('_mean', mean), ('_std', std)):
        if v is not None:
            small.append((name, torch.as_tensor(v, dtype=torch.float32).reshape(-1).cpu()))
    # ---- layout of the staging buffer (16-byte aligned segments)
    segs, off = [], 0
    for k, parts, dim in plan:
        shape = list(parts[0].shape)
        shape[dim] = sum(p.shape[dim] for p in parts)
        nbytes = int(np.prod(shape)) * parts[0].element_size()
        segs.append((k, parts, dim, shape, parts[0].dtype, off, nbytes))
        off = (off + nbytes + 15) // 16 * 16
    for name, v in small:
        segs.append((name, [v], 0, list(v.shape), v.dtype, off, v.numel() * v.element_size()))
        off = (off + v.numel() * v.element_size() + 15) // 16 * 16
    stage, turn = _stager.get(off, device.type == 'cuda')
    for k, parts, dim, shape, dtype, o, nbytes in segs:
        if nbytes:
            dst = stage[o:o + nbytes].view(dtype).view(shape)
            if dim == 1 and len(shape) == 2:
                # edge_index [2, E_i] side by side: row by row, each a contiguous run of the staging buffer (torch.cat along dim 1 into
                # a 2-row destination walks 2 x 32 strided pieces through its generic path: 9 ms per 32-graph batch, 90 % of the collate)
                for r in range(shape[0]):
                    torch.cat([p[r] for p in parts], dim=0, out=dst[r])
            else:
                torch.cat(parts, dim=dim, out=dst)
    dbuf = stage.to(device, non_blocking=True) if device.type != 'cpu' else stage.clone()    # the ONE host-to-device copy
    _stager.sent(turn, device)
    dv = {k: dbuf[o:o + nbytes].view(dtype).view(shape) for k, _, _, shape, dtype, o, nbytes in segs}
    out = Batch()
    for k in first.keys:
        if k.startswith('_'):
            continue
        if k in dv:
            out[k] = dv[k]
        elif not torch.is_tensor(first[k]):
            out[k] = [d[k] for d in data_list]
    batch_vec = torch.empty(n, dtype=torch.int64, device=device)
    x = dv.get('x')
    if x is None:
        raise ValueError('device collate needs node features x')
    K.collate(x, dv.get('_mean'), dv.get('_std'), dv['_gptr'], B, batch_vec, dv.get('edge_index'), dv.get('_eptr'))
    if spatial:
        if knn is None or 'pos' not in dv:
            raise ValueError('spatial=True re-lists the nodes before the k-NN construction: it needs knn= and pos')
        cell = torch.floor(dv['pos'][:, :2].to(torch.float64) / float(knn[0])).to(torch.int64)
        cell = cell - cell.min(0)[0]
        ncx = int(cell[:, 0].max()) + 1 if n else 1
        ncy = int(cell[:, 1].max()) + 1 if n else 1
        key = (batch_vec * ncy + cell[:, 1]) * ncx + cell[:, 0]          # graph-major, then grid row, then grid column
        perm = torch.argsort(key, stable=True)
        for k_ in list(out.keys):
            v = out[k_]
            if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n and k_ not in ('y', 'patch_idx', 'edge_index'):
                out[k_] = v[perm]
        dv['pos'] = out.pos
        out.node_perm = perm                                             # new position i holds the item's node perm[i]
    if knn is not None:
        radius, kmax = knn
        out.edge_index = K.radius_knn(dv['pos'][:, :2], dv['_gptr'], B, float(radius), int(kmax), True)
    out.batch = batch_vec
    out.num_graphs = B
    out._node_counts = counts
    out._gptr = dv['_gptr']
    if has_edges and knn is None:                      # (see the host collate: the edge list is grouped by graph)
        out._eptr = dv['_eptr']
        out._etotal, out._emax = int(sum(ecounts)), int(max(ecounts))
    if spatial or all(getattr(d, '_spatial', False) for d in data_list):
        out._spatial = True
    return out


def _identity_collate(items):
    return items


class DataListLoader(torch.utils.data.DataLoader):
    """Yields python lists of ``Data`` (train.py:52 ``Batch.from_data_list(data)`` expects that)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        kwargs.pop('collate_fn', None)
        super().__init__(dataset, batch_size, shuffle, collate_fn=_identity_collate, **kwargs)


def radius_graph(pos, r, batch=None, loop=False, max_num_neighbors=32):
    """k-nearest (k = max_num_neighbors, + self) within radius r; rows ascending.

    Semantics of torch_cluster 1.4.2's CPU path (cKDTree.query with
    distance_upper_bound = r + 1e-8), SURVEY B.5.  Returns int64 [2, nnz] with
    row = query / aggregating centre and col = neighbour.
    """
    if pos.is_cuda:                     # GPU construction (csrc/knn.hip): any number of graphs per call
        from . import kernels
        n = pos.shape[0]
        if batch is None:
            gptr, B = torch.tensor([0, n], dtype=torch.int32, device=pos.device), 1
        else:                           # ``batch`` sorted ascending, as in every PyG Batch
            B = int(batch[-1].item()) + 1 if n > 0 else 1
            counts = torch.bincount(batch, minlength=B)
            gptr = torch.zeros(B + 1, dtype=torch.int32, device=pos.device)
            gptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        return kernels.get().radius_knn(pos[:, :2], gptr, B, float(r), int(max_num_neighbors), bool(loop))
    if batch is not None:
        raise NotImplementedError('host path: per-graph construction only (as the reference calls it: batch=None)')
    p = pos.detach().cpu().numpy().astype(np.float64)
    tree = cKDTree(p)
    k = max_num_neighbors + 1
    _, col = tree.query(p, k=k, distance_upper_bound=r + 1e-8)
    col = col.reshape(p.shape[0], k)
    row = np.repeat(np.arange(p.shape[0]), k).reshape(p.shape[0], k)
    keep = col < tree.n
    if not loop:
        keep &= col != row
    return torch.from_numpy(np.stack([row[keep], col[keep]]).astype(np.int64))


def spatial_order(pos, cell=100.0):
    """Permutation that lists the nodes of ONE graph cell by cell of a uniform grid (row-major, cell edge = the k-NN radius):
    the neighbours of a node (within ``cell`` pixels) then sit within about three grid rows of it in memory.  The network is
    invariant to the node order (tests/test_full_size_properties_gpu.py); the order only decides how far apart in HBM the rows
    are that the wide neighbour aggregation A*S gathers together -- for graphs of more than ~4000 nodes an arbitrary order
    makes the working set of a (graph, column tile) exceed an XCD's 4 MiB L2 (DESIGN.md, K4 at C5)."""
    p = pos.detach().cpu().numpy() if torch.is_tensor(pos) else np.asarray(pos)
    cx, cy = np.floor(p[:, 0] / cell).astype(np.int64), np.floor(p[:, 1] / cell).astype(np.int64)
    cx, cy = cx - cx.min(initial=0), cy - cy.min(initial=0)
    return torch.from_numpy(np.lexsort((p[:, 0], cx, cy)))      # by grid row, then grid column, then x inside the cell


def reorder_nodes(data, perm):
    """The same graph with its nodes listed in the order ``perm`` (every node-sized tensor is permuted, ``edge_index`` is
    re-labelled and its columns re-sorted by centre so that rows stay ascending as radius_graph emits them)."""
    n = data.num_nodes
    out = data.__class__()
    inv = torch.empty(n, dtype=torch.long)
    inv[perm] = torch.arange(n)
    for k, v in data._all_items():
        if k == 'edge_index':
            ei = inv[v]
            order = torch.argsort(ei[0] * n + ei[1])
            out[k] = ei[:, order]
        elif torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n and k not in ('y', 'patch_idx'):
            out[k] = v[perm]
        else:
            out[k] = v
    return out


class SyntheticCellGraphs(torch.utils.data.Dataset):
    """Seeded synthetic cell graphs.

    graph g (seed = base_seed + g): N_g ~ U{0.8 N .. 1.2 N}; positions uniform in a square of side
    sqrt(N_g * 1784) px (real nucleus density); edges = <= 8 nearest within 100 px + self loop;
    x ~ N(0,1)^F (stands for z-scored features, dataflow/data.py:353); y ~ U{0..classes-1}.
    ``fuse_from`` > 0 draws that many candidate nuclei first and keeps N_g of them with the
    reference's 'fuse' sampler (70 % farthest-point + 30 % random, dataflow/data.py:210-219).
    ``spatial`` lists the nuclei grid cell by grid cell (``spatial_order``) instead of in draw order.
    """

    def __init__(self, num_graphs, mean_nodes, num_features=16, num_classes=3, base_seed=0,
                 radius=100.0, max_neighbours=8, density_px2=1784.0, fuse_from=0, spatial=False):
        self.spatial = spatial
        self.num_graphs, self.mean_nodes, self.num_features = num_graphs, mean_nodes, num_features
        self.num_classes, self.base_seed, self.radius = num_classes, base_seed, radius
        self.max_neighbours, self.density_px2, self.fuse_from = max_neighbours, density_px2, fuse_from
        self.idxlist = ['synthetic_%06d.pt' % i for i in range(num_graphs)]
        self.epoch = self.val_epoch = 0

    def set_epoch(self, epoch):       # train.py:173
        self.epoch = epoch

    def set_val_epoch(self, epoch):   # train.py:36
        self.val_epoch = epoch

    def __len__(self):
        return self.num_graphs

    def __getitem__(self, idx):
        rng = np.random.RandomState(self.base_seed + idx)
        lo, hi = int(round(0.8 * self.mean_nodes)), int(round(1.2 * self.mean_nodes))
        n = int(rng.randint(lo, hi + 1))
        if self.fuse_from > 0:
            cand = max(self.fuse_from, n)
            side = np.sqrt(cand * self.density_px2 / 2.0)  # candidates are 2x denser; sampling restores density
            allpos = rng.uniform(0.0, side, size=(cand, 2))
            keep = fuse_sample(allpos, n, rng)
            pos = allpos[keep]
        else:
            side = np.sqrt(n * self.density_px2)
            pos = rng.uniform(0.0, side, size=(n, 2))
        if self.spatial:
            pos = pos[spatial_order(pos, self.radius).numpy()]
        pos = torch.from_numpy(pos.astype(np.float32))
        x = torch.from_numpy(rng.standard_normal((n, self.num_features)).astype(np.float32))
        y = torch.tensor([int(rng.randint(0, self.num_classes))], dtype=torch.long)
        edge_index = radius_graph(pos, self.radius, None, True, self.max_neighbours)
        d = Data(x=x, pos=pos, y=y, edge_index=edge_index, patch_idx=torch.tensor([idx]))
        if self.spatial:
            d._spatial = True
        return d


def fuse_sample(pos, k, rng, farthest_frac=0.7):
    """'fuse' node sampler: 70 % farthest-point + 30 % uniform random from the rest.

    dataflow/data.py:210-219 with common/utils.py:187-203 (FarthestSampler on the distance table);
    here distances come from coordinates directly.
    """
    n = pos.shape[0]
    kf = int(k * farthest_frac)
    chosen = np.empty(kf, dtype=np.int64)
    dist = np.full(n, np.inf)
    cur = int(rng.randint(n))
    for i in range(kf):
        chosen[i] = cur
        d = ((pos - pos[cur]) ** 2).sum(1)
        dist = np.minimum(dist, d)
        cur = int(dist.argmax())
    rest = np.setdiff1d(np.arange(n), chosen)
    extra = rng.choice(rest, size=k - kf, replace=False)
    return np.sort(np.concatenate([chosen, extra]))


def sample_nodes_batch(pos, counts, ratio, method='fuse', generator=None, farthest_frac=0.7, start=None,
                       distance='coords', rng='torch', order='ascending'):
    """Node sub-sampling for a batch of graphs on the device (dataflow/data.py:195-225): per graph keep
    ``int(n_g * ratio)`` nodes -- 'farthest': farthest-point sampling; 'fuse': 70 % farthest-point + 30 % uniform random
    from the rest; 'random': uniform.  ``pos`` [n,2] float32 on the GPU (graphs concatenated), ``counts`` python list of
    nodes per graph.  Returns (indices int64, list k_g).  Farthest-point picks run in csrc/fps.hip from the coordinates (the
    reference reads rows of an n x n int16 distance table); the first pick of a graph is drawn at random as in the reference
    (``start`` fixes it, for tests).

    ``distance``: 'coords' = exact squared distances (fp64); 'int16' = the entries of the reference's distance table,
    int16(sqrt(dx^2+dy^2)) in float32 (dataflow/construct_feature_graph.py:17-24), recomputed on the fly: the farthest-point
    picks are then index for index those of ``FarthestSampler`` (common/utils.py:187-197) on the stored table.
    ``rng``: 'torch' = the random part is drawn on the device (``generator``); 'python' = as the reference does,
    ``random.sample(remaining indices in ascending order, k)`` on the host (dataflow/data.py:213-215; numpy's
    ``np.random.choice`` for method 'random', :221) -- seed ``random`` / ``numpy.random`` to reproduce its picks.
    ``order``: 'ascending' = indices sorted within each graph; 'reference' = the reference's order per graph, farthest
    picks in pick order followed by the random picks in draw order (``np.concatenate((far_indice, rand_indice))``, :217)."""
    from . import kernels
    K = kernels.get()
    assert distance in ('coords', 'int16') and rng in ('torch', 'python') and order in ('ascending', 'reference')
    dev, B = pos.device, len(counts)
    ks = [int(c * ratio) for c in counts]
    kf = [int(k * farthest_frac) if method == 'fuse' else (k if method == 'farthest' else 0) for k in ks]
    gptr_h = np.concatenate([[0], np.cumsum(counts)])
    gptr = torch.tensor(gptr_h, dtype=torch.int32, device=dev)
    far = None
    if sum(kf) > 0:
        if start is None:
            if rng == 'python':          # FarthestSampler: farthest_pts[0] = np.random.randint(arr.shape[0])
                start = [int(np.random.randint(max(c, 1))) for c in counts]
            else:
                start = [int(torch.randint(max(c, 1), (1,), generator=generator)) for c in counts]
        optr_h = np.concatenate([[0], np.cumsum(kf)])
        optr = torch.tensor(optr_h, dtype=torch.int32, device=dev)
        far = torch.empty(sum(kf), dtype=torch.int32, device=dev)
        K.farthest_point_sample(pos[:, :2].to(torch.float32).contiguous(), gptr, B, max(counts),
                                torch.tensor(start, dtype=torch.int32, device=dev), optr, far, distance == 'int16')
    need = [k - f for k, f in zip(ks, kf)]
    if rng == 'python' or order == 'reference':
        # host-side composition, graph by graph, exactly as dataflow/data.py:205-221 composes ``indice``
        import random as pyrandom
        far_h = far.cpu().numpy().astype(np.int64) if far is not None else np.zeros(0, dtype=np.int64)
        out = []
        for g in range(B):
            lo, n_g = int(gptr_h[g]), int(counts[g])
            fg = far_h[optr_h[g]:optr_h[g + 1]] - lo if far is not None else np.zeros(0, dtype=np.int64)
            if need[g] > 0:
                if method == 'random':
                    rg = np.random.choice(n_g, need[g], replace=False) if rng == 'python' else \
                        torch.randperm(n_g, generator=generator)[:need[g]].numpy()
                else:
                    taken = set(fg.tolist())
                    remain = [i for i in range(n_g) if i not in taken]              # filter_sampled_indice, common/utils.py:200-203
                    if rng == 'python':
                        rg = np.asarray(pyrandom.sample(remain, need[g]), dtype=np.int64)
                    else:
                        rg = np.asarray(remain, dtype=np.int64)[torch.randperm(len(remain), generator=generator)[:need[g]].numpy()]
                ind = np.concatenate((fg, rg.astype(np.int64)), 0)
            else:
                ind = fg
            out.append((np.sort(ind) if order == 'ascending' else ind) + lo)
        return torch.from_numpy(np.concatenate(out) if out else np.zeros(0, dtype=np.int64)).to(dev), ks
    gid = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor(counts, device=dev))
    taken = torch.zeros(int(gptr_h[-1]), dtype=torch.bool, device=dev)
    if far is not None:
        taken[far.long()] = True
    if sum(need) > 0:
        # uniform without replacement from the rest: random keys, nodes already taken pushed to the end of their graph
        key = torch.rand(taken.shape[0], device=dev, generator=generator) + taken.to(torch.float32) * 2.0
        order_ = torch.argsort(gid.to(torch.float32) * 4.0 + key)         # graph-major, random within a graph
        rank = torch.arange(taken.shape[0], device=dev) - gptr[:-1].long()[gid[order_]]
        pick = order_[rank < torch.tensor(need, device=dev)[gid[order_]]]
        taken[pick] = True
    return torch.nonzero(taken).squeeze(1), ks


def partition_by_nodes(data_list, num_parts):
    """Contiguous split of a list of Data into <= num_parts chunks balanced by cumulative node count.

    The rule of torch_geometric.nn.DataParallel.scatter: a graph goes to chunk
    floor(num_parts * m / total) where m is the midpoint of its interval on the cumulative
    node-count axis; empty chunks are dropped (so fewer than num_parts chunks may come back).
    """
    num_parts = min(num_parts, len(data_list))
    counts = torch.tensor([d.num_nodes for d in data_list], dtype=torch.float64)
    cum = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    mid = (cum[:-1] + cum[1:]) / 2.0
    part = (num_parts * mid / cum[-1].item()).long().clamp(0, num_parts - 1)
    chunks = [[] for _ in range(num_parts)]
    for d, p in zip(data_list, part.tolist()):
        chunks[p].append(d)
    return [c for c in chunks if c]
