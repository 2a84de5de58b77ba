"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['tiny_plain', 'tiny_shipped', 'tiny_elu', 'tiny_gin', 'tiny_leaky', 'tiny_tuple', 'medium_plain', 'medium_shipped']


class Bag(object):
    pass


def load_case(name, device='cpu'):
    """(cfg, model input, state_dict, outputs, gradients, state_dict after 3 Adam steps) of tests/golden/<name>.npz.  The model input
    is a Batch-like object (.x, .edge_index, .batch, .y) or, for a case stored in the reference's dense tuple form
    (model/network.py:253-256; cfg['load_data_sparse'] = False), the tuple (x [B,N,F], adj [B,N,N], num_nodes [B], label [B])."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = json.loads(str(z['cfg']))
    if 'in/adj_dense' in z.files:
        batch = tuple(torch.from_numpy(z[k]).to(device) for k in ('in/x_dense', 'in/adj_dense', 'in/counts', 'in/y'))
    else:
        batch = Bag()
        batch.x = torch.from_numpy(z['in/x']).to(device)
        batch.edge_index = torch.from_numpy(z['in/edge_index']).to(device)
        batch.batch = torch.from_numpy(z['in/batch']).to(device)
        batch.y = torch.from_numpy(z['in/y']).to(device)
        batch._node_counts = np.bincount(z['in/batch']).tolist()
    group = lambda p: {k[len(p):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(p)}
    return cfg, batch, group('sd/'), group('out/'), group('grad/'), group('sd3/')


def dense_inputs(batch, dtype=torch.float64):
    """A case's input in the dense tuple form, in ``dtype``: as stored, or densified with the oracle's to_dense_adj / to_dense_batch."""
    from oracle import dense_ref
    if isinstance(batch, tuple):
        return (batch[0].to(dtype), batch[1].to(dtype).clone(), batch[2], batch[3].view(-1))
    adj = dense_ref.to_dense_adj(batch.edge_index, batch.batch)
    xd, counts = dense_ref.to_dense_batch(batch.x, batch.batch)
    return (xd.to(dtype), adj.to(dtype), counts, batch.y.view(-1))


def build_model(cls, cfg, **over):
    kw = dict(cfg)
    kw.update(over)
    return cls(kw['max_num_nodes'], kw['input_dim'], kw['hidden_dim'], kw['embedding_dim'], True, True,
               kw['hidden_dim'], 3, kw['assign_ratio'], [50], concat=True, gcn_name=kw.get('gcn_name', 'SAGE'),
               collect_assign=kw.get('collect_assign', False), load_data_sparse=kw.get('load_data_sparse', True),
               norm_adj=kw.get('norm_adj', False), activation=kw.get('activation', 'relu'),
               drop_out=kw.get('drop_out', 0.), jk=kw.get('jk', False))


def rel_err(a, b, atol=1e-7):
    """max|a-b| / (max|b| + atol/1e-4): the 'within 1e-4 relative, fp32' yardstick of the north star.
    The atol term only matters for tensors that are mathematically zero (e.g. the gradient of the
    attention bias, to which the softmax is invariant): there 1e-4 * rel_err is an absolute error."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (float(b.abs().max()) + atol / 1e-4))


def elementwise_excess(a, b, rtol=1e-4, atol=None):
    """Elementwise yardstick beside the max-norm one: the largest value of |a-b| / (rtol*|b| + atol) over all elements
    (<= 1 means every element satisfies |a-b| <= rtol*|b| + atol).  ``atol`` defaults to rtol * rms(b) + 1e-7: the rounding
    error of an fp32 sum of mixed-sign terms scales with the magnitude of the TERMS, not with the (possibly cancelled)
    result, so elements near zero are held to rtol times the tensor's typical magnitude; that is still up to
    max|b|/rms(b) times tighter for them than the max-norm bound."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    if atol is None:
        atol = rtol * float(b.pow(2).mean().sqrt()) + 1e-7
    return float(((a - b).abs() / (rtol * b.abs() + atol)).max())
