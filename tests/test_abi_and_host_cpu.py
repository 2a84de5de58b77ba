"""CPU: the C-ABI library builds/loads and exports every declared symbol; host-side containers behave like the
torch_geometric objects the reference feeds its model with; the product refuses to compute without the HIP path."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import _abi, kernels, network
from cgc_net_amd.data import Batch, Data, DataListLoader, SyntheticCellGraphs, partition_by_nodes, radius_graph
from oracle import dense_ref
from util import build_model, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    header = open(os.path.join(ROOT, 'include', 'cgc_hip.h')).read()
    declared = set(re.findall(r'^(?:int|int64_t|void\*) (cgc_\w+)\(', header, flags=re.M))
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)
    lib = ctypes.CDLL(kernels.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cgc_abi_version() == _abi.ABI_VERSION == int(re.search(r'#define CGC_ABI_VERSION (\d+)', header).group(1))


@pytest.mark.skipif(torch.cuda.is_available(), reason='only meaningful on a host without a GPU')
def test_no_cpu_fallback():
    cfg, batch, sd, *_ = load_case('tiny_plain')
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        model(batch)


def test_state_dict_keys_match_reference_layout():
    for name in ('medium_plain', 'medium_shipped'):
        cfg, _, sd, *_ = load_case(name)
        model = build_model(network.SoftPoolingGcnEncoder, cfg)
        assert list(model.state_dict().keys()) == list(sd.keys())
        for k, v in model.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), k
    m = network.SoftPoolingGcnEncoder(600, 16, 20, 20, True, True, 20, 3, 0.1, [50], drop_out=0.2)
    assert 'pred_model.3.weight' in m.state_dict()          # Dropout shifts the last Linear


def test_radius_graph_semantics():
    rng = np.random.RandomState(0)
    pos = torch.from_numpy(rng.uniform(0, 500, size=(200, 2)).astype(np.float32))
    ei = radius_graph(pos, 100.0, None, True, 8)
    row, col = ei.numpy()
    assert (np.diff(row) >= 0).all()                         # rows ascending
    d = np.linalg.norm(pos.numpy()[row] - pos.numpy()[col], axis=1)
    assert (d <= 100.0 + 1e-4).all()
    cnt = np.bincount(row, minlength=200)
    assert cnt.max() <= 9 and (row == col).sum() == 200      # <= 8 neighbours + the self loop
    # brute force: the kept neighbours are the nearest ones
    full = np.linalg.norm(pos.numpy()[:, None] - pos.numpy()[None], axis=2)
    for i in (0, 17, 199):
        want = np.sort(full[i][full[i] <= 100.0])[:9]
        got = np.sort(d[row == i])
        assert np.allclose(got, want, atol=1e-3)


def test_batch_collate_and_loader(torch_kernels):
    ds = SyntheticCellGraphs(6, 40, num_features=5, base_seed=3)
    assert ds[2].x.equal(ds[2].x) and ds[2].x.equal(SyntheticCellGraphs(6, 40, 5, base_seed=3)[2].x)   # seeded
    items = [ds[i] for i in range(3)]
    b = Batch.from_data_list(items)
    n = [d.num_nodes for d in items]
    assert b.x.shape[0] == sum(n) and b.batch.tolist() == sum(([g] * k for g, k in enumerate(n)), [])
    off = np.cumsum([0] + n)
    for g, d in enumerate(items):
        sel = (b.batch[b.edge_index[0]] == g)
        assert torch.equal(b.edge_index[:, sel] - int(off[g]), d.edge_index)
    # the notes both collates leave for the graph-by-graph CSR build (cgc_graph_build_local): the edge list is grouped by graph
    ec = [int(d.edge_index.shape[1]) for d in items]
    assert b._eptr.dtype == torch.int32 and b._eptr.tolist() == [0] + list(np.cumsum(ec)) and b._etotal == sum(ec) and b._emax == max(ec)
    for g in range(3):
        seg = b.edge_index[:, b._eptr[g]:b._eptr[g + 1]]
        assert int(seg.min()) >= off[g] and int(seg.max()) < off[g + 1]
    assert '_eptr' not in b.keys and '_gptr' not in b.keys and [k for k, _ in b] == b.keys       # notes are not data ...
    moved = b.to('cpu')
    assert torch.equal(moved._eptr, b._eptr) and moved._etotal == b._etotal and moved._emax == b._emax    # ... but they travel with to()
    devb = Batch.from_data_list(items, device='cpu')
    assert torch.equal(devb._eptr, b._eptr) and devb._etotal == b._etotal and devb._emax == b._emax
    assert not hasattr(Batch.from_data_list([Data(x=d.x, pos=d.pos, y=d.y) for d in items], device='cpu', knn=(100.0, 8)), '_eptr')
    loader = DataListLoader(ds, batch_size=4, shuffle=False)
    first = next(iter(loader))
    assert isinstance(first, list) and len(first) == 4 and isinstance(first[0], Data)
    loader.dataset.set_epoch(3)
    assert loader.dataset.epoch == 3 and len(loader.dataset.idxlist) == 6


def test_partition_by_cumulative_node_count():
    ds = SyntheticCellGraphs(8, 50, num_features=2, base_seed=1)
    items = [ds[i] for i in range(8)]
    chunks = partition_by_nodes(items, 4)
    assert sum(len(c) for c in chunks) == 8 and len(chunks) == 4
    flat = [d for c in chunks for d in c]
    assert all(a is b for a, b in zip(flat, items))          # contiguous, order kept
    sizes = [sum(d.num_nodes for d in c) for c in chunks]
    assert max(sizes) - min(sizes) <= 2 * max(d.num_nodes for d in items)


def test_oracle_matches_golden():
    """The oracle itself against the reference-generated vectors (the pin)."""
    from util import CASES, rel_err
    for name in CASES:
        cfg, batch, sd, out, grad, sd3 = load_case(name)
        m = build_model(dense_ref.SoftPoolingGcnEncoder, cfg)
        m.load_state_dict(sd)
        m.train()
        logits, loss = m(batch)
        loss.backward()
        assert rel_err(logits, out['logits']) < 1e-5 and rel_err(loss, out['loss']) < 1e-5
        for k, p in m.named_parameters():
            assert rel_err(p.grad, grad[k]) < 1e-5, (name, k)


def test_device_collate_front_end_matches_host_collate(torch_kernels):
    """Batch.from_data_list(..., device=): one packed copy + one finishing kernel == the host collate, z-score and per-graph
    radius_graph (kernel entry points through the CPU twin here; the HIP path is tests/test_kernels_gpu.py)."""
    ds = SyntheticCellGraphs(5, 120, 6, base_seed=4)
    items = [ds[i] for i in range(5)]
    mean, std = torch.linspace(-0.5, 0.5, 6), torch.linspace(0.5, 2.0, 6)
    host = Batch.from_data_list([Data(x=(d.x - mean) / std, pos=d.pos, y=d.y, edge_index=d.edge_index) for d in items])
    dev = Batch.from_data_list(items, device='cpu', mean=mean, std=std)
    assert torch.equal(dev.x, host.x) and torch.equal(dev.pos, host.pos) and torch.equal(dev.y, host.y)
    assert torch.equal(dev.batch, host.batch) and torch.equal(dev.edge_index, host.edge_index)
    assert dev._node_counts == host._node_counts and dev.num_graphs == 5
    # edges built at collate time from the positions (items without edge_index)
    bare = [Data(x=d.x, pos=d.pos, y=d.y) for d in items]
    built = Batch.from_data_list(bare, device='cpu', knn=(100.0, 8))
    key = lambda e: sorted(zip(e[0].tolist(), e[1].tolist()))
    assert key(built.edge_index) == key(host.edge_index)
    assert torch.equal(built.x, torch.cat([d.x for d in items]))
    # spatial=True: the same graphs with their nodes re-listed grid cell by grid cell (a permutation inside every graph)
    sp = Batch.from_data_list(bare, device='cpu', knn=(100.0, 8), spatial=True)
    perm = sp.node_perm
    assert torch.equal(torch.sort(perm)[0], torch.arange(perm.numel())) and torch.equal(sp.batch, host.batch)
    assert torch.equal(sp.x, built.x[perm]) and torch.equal(sp.pos, built.pos[perm])
    relabelled = perm[sp.edge_index]                                    # back to the items' node numbering
    assert key(relabelled) == key(host.edge_index)
    cell = torch.floor(sp.pos / 100.0).long()
    for g in range(5):                                                   # grid rows ascend inside every graph
        cy = cell[sp.batch == g, 1]
        assert bool((cy[1:] >= cy[:-1]).all())
    from cgc_net_amd.data import reorder_nodes, spatial_order
    d0 = reorder_nodes(items[0], spatial_order(items[0].pos))
    assert key(spatial_order(items[0].pos)[d0.edge_index]) == key(items[0].edge_index) and bool((d0.edge_index[0][1:] >= d0.edge_index[0][:-1]).all())


# ------------------------------------------------------------------ F4: checkpoint interchange + evaluation protocol
def test_image_level_vote_matches_reference_fixture():
    """evalio.ImageLevelVote against vectors produced by the reference's common/metric.py (tests/golden/make_vote_golden.py)."""
    import json
    from cgc_net_amd import evalio
    cases = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'vote_cases.json')))
    assert len(cases) >= 6
    for c in cases:
        v = evalio.ImageLevelVote(c['ground_truth'])
        half = len(c['patches']) // 2
        for nm, lb in zip(c['patches'][:half], c['labels'][:half]):
            v.patch_result(nm, lb)
        v.batch_patch_result(c['patches'][half:], c['labels'][half:])
        acc, bacc = v.final_result()
        assert abs(acc - c['acc']) < 1e-12 and abs(bacc - c['binary_acc']) < 1e-12


def test_checkpoint_roundtrip_and_reference_key_layout(tmp_path, torch_kernels):
    """The dict of train.py:202-207 through save/load (common/utils.py:82-94); a checkpoint saved from the wrapped model
    (keys prefixed 'module.') loads into a bare model; a reference-generated state_dict (golden fixture) loads strictly."""
    from cgc_net_amd import evalio
    cfg, _, sd, _, _, _ = load_case('tiny_shipped')
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    evalio.load_reference_state(model, sd)                    # the fixture's state_dict was written by the reference's own model
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    state = evalio.checkpoint_state(model, opt, epoch=4, loss=0.5, val_acc=0.75)
    assert set(state) == {'epoch', 'loss', 'state_dict', 'optimizer', 'val_acc'} and state['epoch'] == 5
    f = os.path.join(str(tmp_path), 'run', 'weight.pth.tar')
    evalio.save_checkpoint(state, True, f)
    assert os.path.isfile(os.path.join(str(tmp_path), 'run', 'model_best.pth.tar'))
    ck = evalio.load_checkpoint(f)
    other = build_model(network.SoftPoolingGcnEncoder, cfg)
    with torch.no_grad():
        for p in other.parameters():
            p.add_(1.0)
    wrapped = {'state_dict': {'module.' + k: v for k, v in ck['state_dict'].items()}}
    evalio.load_reference_state(other, wrapped)
    for (k, a), (_, b) in zip(sorted(model.state_dict().items()), sorted(other.state_dict().items())):
        assert torch.equal(a, b), k
    with pytest.raises(ValueError):
        evalio.load_checkpoint(os.path.join(str(tmp_path), 'missing.pth.tar'))


def test_evaluate_protocol(torch_kernels):
    """train.py:21-91: logits averaged over test_time passes, votes collected from every pass."""
    from cgc_net_amd import evalio
    ds = SyntheticCellGraphs(6, 40, 16, base_seed=1)
    ds.idxlist = ['/x/img%d_grade_%d_patch0.pt' % (i // 2, 1 + i // 2) for i in range(6)]

    class WithIdx(torch.utils.data.Dataset):
        idxlist = ds.idxlist

        def __len__(self):
            return 6

        def __getitem__(self, i):
            d = ds[i]
            d.patch_idx = torch.tensor([i])
            return d

        def set_val_epoch(self, e):
            self.epoch = e
    loader = DataListLoader(WithIdx(), batch_size=3)
    torch.manual_seed(0)
    model = network.SoftPoolingGcnEncoder(600, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True)
    vote = evalio.ImageLevelVote(['img0_grade_1', 'img1_grade_2', 'img2_grade_3'])
    res = evalio.evaluate(loader, model, vote, test_time=2)
    assert set(res) == {'patch_acc', 'img_acc', 'binary_acc'} and all(0.0 <= v <= 1.0 for v in res.values())
    assert sum(len(v) for v in vote.prediction.values()) == 12 and loader.dataset.epoch == 1     # 6 patches x 2 passes
    assert model.training                                                                     # mode restored


def test_optimiser_wrapper_on_host_tensors():
    """cgc_net_amd.optim.Adam without the step sequencer behind it (CPU tensors, an arbitrary module passed as ``model``): torch's
    own Adam trajectory through the cached-list path; the one-launch path is never taken (there are no flat gradient buffers)."""
    import torch
    from cgc_net_amd.optim import Adam
    torch.manual_seed(0)
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
    b.load_state_dict(a.state_dict())
    oa = Adam(a.parameters(), lr=1e-2, weight_decay=1e-4, model=a)
    ob = torch.optim.Adam(b.parameters(), lr=1e-2, weight_decay=1e-4)
    x = torch.randn(7, 5)
    for _ in range(4):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).pow(2).sum().backward()
            o.step()
    assert not oa._fast_ready()
    assert torch.allclose(a.weight, b.weight, atol=1e-7) and torch.allclose(a.bias, b.bias, atol=1e-7)
    assert all(float(st['step']) == 4.0 for st in oa.state_dict()['state'].values())


def test_optimiser_wrapper_late_gradients_and_grad_mul_on_host_tensors():
    """(i) a parameter that receives its first gradient on a LATER step must start being updated then, as with torch.optim.Adam (the
    cached parameter lists are rebuilt when the set of parameters with a gradient changes); (ii) grad_mul on the paths that end in
    torch's kernels scales what the update sees and leaves the caller's p.grad untouched."""
    import torch
    from cgc_net_amd.optim import Adam

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)

        def forward(self, x, use_b):
            return self.a(x) + (self.b(x) if use_b else 0.0)
    torch.manual_seed(0)
    m, r = Two(), Two()
    r.load_state_dict(m.state_dict())
    b0 = m.b.weight.detach().clone()
    om = Adam(m.parameters(), lr=1e-2, weight_decay=1e-4, model=m, grad_mul=0.5)
    orr = torch.optim.Adam(r.parameters(), lr=1e-2, weight_decay=1e-4)
    x = torch.randn(6, 4)
    for step in range(6):
        use_b = step >= 2 and step != 4
        for mod, o in ((m, om), (r, orr)):
            o.zero_grad()
            mod(x, use_b).pow(2).sum().backward()
        before = [p.grad.clone() if p.grad is not None else None for p in m.parameters()]
        for p in r.parameters():
            if p.grad is not None:
                p.grad.mul_(0.5)
        om.step()
        orr.step()
        for p, g in zip(m.parameters(), before):
            assert (p.grad is None) == (g is None) and (g is None or torch.equal(p.grad, g))       # the caller's gradients: untouched
    for (k, p), (_, q) in zip(m.state_dict().items(), r.state_dict().items()):
        assert torch.allclose(p, q, atol=1e-7), k
    assert float((m.b.weight.detach() - b0).abs().max()) > 1e-3       # (b did move)


def test_per_pass_gradient_buffer_slices():
    """native._grad_buffer: the head's and the three levels' flat gradient buffers are 256-byte-aligned slices of ONE buffer per
    backward pass (what DataParallel all-reduces in place and cgc_adam_step reads); a slot asked for twice starts a new pass; an owner
    without the four sizes, or an unexpected size, gets a private buffer."""
    import torch
    from cgc_net_amd import native

    class Owner(object):
        pass
    o = Owner()
    o._flat_sizes = {0: 9203, 1: 1500001, 2: 70000, 3: 333}
    dev = torch.device('cpu')
    bufs = [native._grad_buffer(o, s, o._flat_sizes[s], dev) for s in (0, 3, 2, 1)]      # the order backward visits them
    base = o._step_flat
    pad = lambda n: -(-n // 64) * 64
    assert base.numel() == sum(pad(n) for n in o._flat_sizes.values()) and o._step_taken == {0, 1, 2, 3}
    for s, b in zip((0, 3, 2, 1), bufs):
        off = (b.data_ptr() - base.data_ptr()) // 4
        assert off == sum(pad(o._flat_sizes[t]) for t in range(s)) and off % 64 == 0 and b.numel() == o._flat_sizes[s]
    again = native._grad_buffer(o, 0, 9203, dev)                                         # the next backward pass
    assert o._step_flat is not base and again.data_ptr() == o._step_flat.data_ptr() and o._step_taken == {0}
    private = native._grad_buffer(o, 1, 123, dev)                                        # not the registered size
    assert private.numel() == 123 and not (o._step_flat.data_ptr() <= private.data_ptr() < o._step_flat.data_ptr() + 4 * o._step_flat.numel())
    assert native._grad_buffer(None, 0, 10, dev).numel() == 10
