#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/*.npz  (BUILD CONTAINER ONLY).

Imports the REFERENCE implementation (``/root/reference/model/network.py``) under a small
torch_geometric stand-in (tests/golden/pyg_standin, semantics = oracle/dense_ref.py) with
``Tensor.cuda`` patched to the identity (the reference hard-codes ``.cuda()`` at
model/network.py:180), runs it on seeded inputs and stores

  in/*      x, edge_index, batch, y
  cfg       json: constructor arguments
  sd/*      the initial state_dict
  out/*     train-mode logits + loss, the collected assignment matrices
  grad/*    d loss / d parameter for every parameter
  sd3/*     state_dict after 3 Adam(lr 1e-3, wd 1e-4) steps on the same batch (train.py:174-184,
            common/utils.py:119-121) -- includes the BatchNorm running statistics
  out/eval_logits3   eval-mode logits after those steps

While doing so it asserts that oracle/dense_ref.py reproduces the reference bit-for-bit-ish
(<= 2e-6 scaled abs on forward/gradients, <= 1e-4 after the Adam steps) on every stored quantity: that is the pin of the oracle.

Neither the reference nor this script's stand-in travels to the GPU box; only the .npz files do.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, 'pyg_standin'), ROOT]
REF = os.environ.get('CGC_REFERENCE', '/root/reference')
sys.path.append(REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self   # model/network.py:180

from model import network as refnet  # noqa: E402  (the reference)
from oracle import dense_ref  # noqa: E402
import cgc_net_amd  # noqa: E402
from cgc_net_amd.data import Batch, Data, SyntheticCellGraphs, radius_graph  # noqa: E402


def tiny_batch(seed, feat):
    """3 ragged graphs (5, 9, 12 nodes) with awkward structure: a duplicated edge, a node with no
    edges at all (deg 0 -> the clamp(min=1) matters), a node whose only edge is its self loop."""
    rng = np.random.RandomState(seed)
    graphs = []
    for n in (5, 9, 12):
        pos = torch.from_numpy(rng.uniform(0, 120, size=(n, 2)).astype(np.float32))
        ei = radius_graph(pos, 60.0, None, True, 3)
        ei = ei[:, ei[0] != 1]                                   # node 1: no outgoing edges, no self loop
        ei = ei[:, ~((ei[0] == 2) & (ei[1] != 2))]               # node 2: self loop only
        ei = torch.cat([ei, ei[:, :2]], dim=1)                   # duplicate two edges
        x = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32))
        graphs.append(Data(x=x, pos=pos, y=torch.tensor([int(rng.randint(3))]), edge_index=ei))
    return Batch.from_data_list(graphs)


def tiny_dense_tuple(seed, feat, pad_to):
    """The same kind of graphs handed over in the reference's SECOND input form (model/network.py:253-256; the visualisation branch
    of evaluate(), train.py:38-49): (x [B, N, F], adj [B, N, N] 0/1, num_nodes [B], label [B]) with N padded beyond the largest graph,
    as the dense-dict datasets pad to a fixed size (dataflow/data.py:234,268)."""
    b = tiny_batch(seed, feat)
    counts = np.bincount(b.batch.numpy()).tolist()
    B = len(counts)
    x = torch.zeros(B, pad_to, feat)
    adj = torch.zeros(B, pad_to, pad_to)
    off = np.concatenate([[0], np.cumsum(counts)])
    for g in range(B):
        x[g, :counts[g]] = b.x[off[g]:off[g + 1]]
    gi = b.batch[b.edge_index[0]]
    adj[gi, b.edge_index[0] - torch.from_numpy(off)[gi], b.edge_index[1] - torch.from_numpy(off)[gi]] = 1.0
    return (x, adj, torch.tensor(counts), b.y.view(-1))


CASES = {
    # name: (batch builder, ctor kwargs)
    'tiny_plain': (lambda: tiny_batch(1, 4),
                   dict(max_num_nodes=64, input_dim=4, hidden_dim=8, embedding_dim=8, assign_ratio=0.25)),
    'tiny_shipped': (lambda: tiny_batch(2, 4),
                     dict(max_num_nodes=64, input_dim=4, hidden_dim=8, embedding_dim=8, assign_ratio=0.25,
                          norm_adj=True, jk=True)),
    'tiny_elu': (lambda: tiny_batch(3, 4),
                          dict(max_num_nodes=64, input_dim=4, hidden_dim=8, embedding_dim=8, assign_ratio=0.25,
                               activation='elu', norm_adj=True)),
    # round 5: the branches the five cases above do not reach -- gcn_name != 'SAGE' (model/network.py:96-99), leaky ReLU (:90-91),
    # the dense tuple input form (:253-256)
    'tiny_gin': (lambda: tiny_batch(4, 4),
                 dict(max_num_nodes=64, input_dim=4, hidden_dim=8, embedding_dim=8, assign_ratio=0.25, gcn_name='GIN')),
    'tiny_leaky': (lambda: tiny_batch(5, 4),
                   dict(max_num_nodes=64, input_dim=4, hidden_dim=8, embedding_dim=8, assign_ratio=0.25,
                        activation='leakyrelu', norm_adj=True, jk=True)),
    'tiny_tuple': (lambda: tiny_dense_tuple(6, 4, 16),
                   dict(max_num_nodes=64, input_dim=4, hidden_dim=8, embedding_dim=8, assign_ratio=0.25,
                        norm_adj=True, jk=True, load_data_sparse=False)),
    'medium_plain': (lambda: Batch.from_data_list([SyntheticCellGraphs(4, 300, 16, base_seed=11)[i] for i in range(4)]),
                     dict(max_num_nodes=600, input_dim=16, hidden_dim=20, embedding_dim=20, assign_ratio=0.1)),
    'medium_shipped': (lambda: Batch.from_data_list([SyntheticCellGraphs(4, 300, 16, base_seed=23)[i] for i in range(4)]),
                       dict(max_num_nodes=600, input_dim=16, hidden_dim=20, embedding_dim=20, assign_ratio=0.1,
                            norm_adj=True, jk=True)),
}


def build(cls, kw):
    # positional layout of train.py:254-261: (maxn, in, hidden, out, bias, bn, assign_hidden, classes, ratio, [50])
    return cls(kw['max_num_nodes'], kw['input_dim'], kw['hidden_dim'], kw['embedding_dim'],
               True, True, kw['hidden_dim'], 3, kw['assign_ratio'], [50], concat=True, gcn_name=kw.get('gcn_name', 'SAGE'),
               collect_assign=True, load_data_sparse=kw.get('load_data_sparse', True), norm_adj=kw.get('norm_adj', False),
               activation=kw.get('activation', 'relu'), drop_out=0., jk=kw.get('jk', False))


def run(model, batch_in, steps=3):
    """train-mode fwd/bwd, then ``steps`` Adam steps, then an eval forward."""
    out = {}
    if isinstance(batch_in, tuple):        # (_re_norm_adj writes the diagonal of its input in place, model/network.py:186: fresh copies)
        class _Fresh(object):
            def __iter__(self):
                return iter(tuple(t.clone() for t in batch_in))

            def __getitem__(self, i):
                return batch_in[i].clone()
        batch = _Fresh()
    else:
        batch = batch_in
    model.train()
    model.zero_grad()
    logits, loss = model(batch)
    out['out/logits'], out['out/loss'] = logits.detach().numpy().copy(), loss.detach().numpy().copy()
    for i, s in enumerate(model.assign_matrix):
        out['out/assign%d' % (i + 1)] = s.numpy().copy()
    loss.backward()
    for k, p in model.named_parameters():
        out['grad/' + k] = p.grad.detach().numpy().copy()
    # the first of the three steps re-uses the very same batch (BN buffers were already updated once
    # by the forward above; the fixture records exactly this sequence: 1 fwd/bwd + 3 full steps).
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(steps):
        _, l = model(batch)
        l = torch.mean(l)
        opt.zero_grad()
        l.backward()
        opt.step()
    for k, v in model.state_dict().items():
        out['sd3/' + k] = v.detach().numpy().copy()
    model.eval()
    with torch.no_grad():
        out['out/eval_logits3'] = model(batch).numpy().copy()
    return out


def main():
    worst = {}
    for name, (mk, kw) in CASES.items():
        batch = mk()
        torch.manual_seed(1234)
        ref = build(refnet.SoftPoolingGcnEncoder, kw)
        # de-trivialise BN affine parameters so that their gradients are exercised
        with torch.no_grad():
            for k, p in ref.named_parameters():
                if '.bn' in k:
                    p.add_(0.1 * torch.randn_like(p))
        sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
        ora = build(dense_ref.SoftPoolingGcnEncoder, kw)
        missing = ora.load_state_dict(sd0, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        got_ref, got_ora = run(ref, batch), run(ora, batch)
        for k in got_ref:
            a, b = got_ref[k].astype(np.float64), got_ora[k].astype(np.float64)
            err = np.abs(a - b).max() / max(1.0, np.abs(a).max())
            # Adam divides by sqrt(v): tiny gradient differences are amplified in the stepped weights
            tol = 1e-4 if (k.startswith('sd3/') or k == 'out/eval_logits3') else 2e-6
            worst[k.split('/')[0]] = max(worst.get(k.split('/')[0], 0.0), err)
            assert err <= tol, 'oracle != reference on %s/%s: %g' % (name, k, err)
        if isinstance(batch, tuple):
            fix = {'cfg': np.array(json.dumps(kw)), 'in/x_dense': batch[0].numpy(), 'in/adj_dense': batch[1].numpy(),
                   'in/counts': batch[2].numpy(), 'in/y': batch[3].numpy()}
        else:
            fix = {'cfg': np.array(json.dumps(kw)),
                   'in/x': batch.x.numpy(), 'in/edge_index': batch.edge_index.numpy(),
                   'in/batch': batch.batch.numpy(), 'in/y': batch.y.numpy()}
        fix.update({'sd/' + k: v.numpy() for k, v in sd0.items()})
        fix.update(got_ref)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **fix)
        print('%-20s nodes=%d edges=%d params=%d  loss=%.6f  -> %s (%.1f KB)' % (
            name, int(batch[2].sum()) if isinstance(batch, tuple) else batch.x.shape[0],
            int(batch[1].sum()) if isinstance(batch, tuple) else batch.edge_index.shape[1],
            sum(p.numel() for p in ref.parameters()), float(got_ref['out/loss']),
            os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))
    print('oracle vs reference, worst scaled abs error per group:', {k: float('%.3g' % v) for k, v in worst.items()})


if __name__ == '__main__':
    main()
