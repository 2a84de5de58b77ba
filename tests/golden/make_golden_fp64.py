#!/usr/bin/env python
"""Generate tests/golden/<case>_fp64.npz  (BUILD CONTAINER ONLY): the REFERENCE evaluated in float64.

The fp32 fixtures (make_golden.py) carry the reference's own fp32 gradients; a gradient bar of 1e-4 cannot be held against
those, because the reference's fp32 arithmetic is itself 2e-5 .. 6e-4 away from the exact value on some parameters and takes
ReLU signs / max-readout winners that another fp32 evaluation may take differently (tests/discrete.py).  This script pins the
yardstick of that bar to data the REFERENCE produced: it imports ``/root/reference/model/network.py`` (same stand-in package as
make_golden.py), casts the module to float64 and runs it through its dense tuple input form
``(x[B,N,F], adj[B,N,N], num_nodes, label)`` (model/network.py:253-256) -- adjacency from the reference's own
``to_dense_adj`` (model/utils.py:3-36) -- on the inputs and the initial ``state_dict`` stored in ``<case>.npz``.  Stored:

  out64/logits, out64/loss        float64 forward
  grad64/<parameter>              float64 d loss / d parameter
  pre64/<block>.gcn<k>            the value each block activation is applied to (model/network.py:114-116: the convolution
                                  output, [B, N, F] in the dense layout) -- recorded by a forward-pre-hook on ``active<k>``
  embed64/<level>                 the tensor each readout maximises over (model/network.py:264,275,284), [B, N, D]
  win64/<level>                   its argmax over the node axis, [B, D]
  ulp64/<parameter>               how far the reference's OWN float64 gradient of that parameter moves (max-norm, relative) when every
                                  parameter is perturbed by one float32 rounding (relative 2^-24, random signs; worst of 8 draws): what
                                  no float32 evaluation of this network can be expected to beat on this input.  Printed next to the
                                  measured error by tests/discrete.py::compare_with_reference_fp64, whose bar is the plain 1e-4 on
                                  every parameter (round 4's max(1e-4, 2 x this) widening was never needed and is gone)

and asserts oracle/dense_ref.py (float64) == reference (float64) to 1e-11 on all of them.  Only the .npz files travel.
"""
import copy
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, 'pyg_standin'), ROOT, os.path.join(ROOT, 'tests')]
REF = os.environ.get('CGC_REFERENCE', '/root/reference')
sys.path.append(REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self   # model/network.py:180

from model import network as refnet  # noqa: E402  (the reference)
from model.utils import to_dense_adj as ref_to_dense_adj  # noqa: E402
from torch_geometric.utils import to_dense_batch  # noqa: E402  (stand-in; PyG 1.2.1 semantics: returns counts)
from oracle import dense_ref  # noqa: E402
from make_golden import build  # noqa: E402
from util import CASES, load_case  # noqa: E402


def run_reference64(ref, inp):
    pre, embeds, hooks = {}, {}, []
    for bname, blk in ref.named_children():
        if not hasattr(blk, 'gcn1'):
            continue
        for k in (1, 2, 3):
            def rec(mod, args, name='%s.gcn%d' % (bname, k)):
                pre[name] = args[0].detach().clone()          # (the activation is in place: copy before it runs)
            hooks.append(getattr(blk, 'active%d' % k).register_forward_pre_hook(rec))
    for level in (1, 2, 3):
        src = getattr(ref, 'jk%d' % level) if ref.jk else getattr(ref, 'GCN_embed_%d' % level)

        def rec_out(mod, args, out, level=level):
            embeds[level] = out.detach().clone()
        hooks.append(src.register_forward_hook(rec_out))
    try:
        ref.train()
        ref.zero_grad()
        logits, loss = ref(inp)
        loss.backward()
    finally:
        for h in hooks:
            h.remove()
    return logits.detach(), loss.detach(), pre, embeds


def main():
    worst = 0.0
    for name in CASES:
        cfg, batch, sd, _out, _grad, _sd3 = load_case(name)
        ref = build(refnet.SoftPoolingGcnEncoder, cfg)
        ref.load_state_dict(sd)
        ref = ref.double()
        ref.load_data_sparse = False
        if isinstance(batch, tuple):                       # a case stored in the dense tuple form already (tiny_tuple)
            x, adj, counts, y = batch[0], batch[1].double(), batch[2], batch[3].view(-1)
        else:
            adj = ref_to_dense_adj(batch.edge_index, batch.batch).double()
            x, counts = to_dense_batch(batch.x, batch.batch)
            y = batch.y.view(-1)
        inp = (x.double(), adj.clone(), counts, y)
        logits, loss, pre, embeds = run_reference64(ref, inp)
        assert logits.dtype == torch.float64 and loss.dtype == torch.float64
        fix = {'out64/logits': logits.numpy(), 'out64/loss': loss.numpy(), 'counts': np.asarray([int(c) for c in counts])}
        for k, p in ref.named_parameters():
            assert p.grad.dtype == torch.float64
            fix['grad64/' + k] = p.grad.numpy().copy()
        for k, v in pre.items():
            fix['pre64/' + k] = v.numpy()
        for level, e in embeds.items():
            fix['embed64/%d' % level] = e.numpy()
            fix['win64/%d' % level] = e.max(dim=1)[1].numpy()
        # conditioning of every parameter gradient: the reference in float64 on parameters moved by one float32 rounding
        g0 = {k: p.grad.clone() for k, p in ref.named_parameters()}
        sens = {k: 0.0 for k in g0}
        gen = torch.Generator().manual_seed(99)
        for _draw in range(8):
            pert = build(refnet.SoftPoolingGcnEncoder, cfg)
            sdp = {}
            for k, v in sd.items():
                if v.dtype.is_floating_point:
                    sign = torch.randint(0, 2, v.shape, generator=gen).double() * 2 - 1
                    sdp[k] = v.double() * (1.0 + sign * 2.0 ** -24)
                else:
                    sdp[k] = v
            pert = pert.double()
            pert.load_state_dict(sdp)
            pert.load_data_sparse = False
            pert.train()
            pert.zero_grad()
            _, lp = pert((x.double(), adj.clone(), counts, y))
            lp.backward()
            for k, p in pert.named_parameters():
                m = float(g0[k].abs().max())
                if m > 1e-12:
                    sens[k] = max(sens[k], float((p.grad - g0[k]).abs().max()) / m)
        for k, v in sens.items():
            fix['ulp64/' + k] = np.float64(v)
        # the oracle in float64 must be the same function
        ora = build(dense_ref.SoftPoolingGcnEncoder, cfg)
        ora.load_state_dict(sd)
        ora = ora.double()
        ora.load_data_sparse = False
        ora.train()
        ol, oloss = ora((x.double(), adj.clone(), counts, y))
        oloss.backward()
        errs = [float((ol - logits).abs().max() / logits.abs().max()), float((oloss - loss).abs() / loss.abs())]
        og = dict(ora.named_parameters())
        for k, p in ref.named_parameters():
            errs.append(float((og[k].grad - p.grad).abs().max() / max(float(p.grad.abs().max()), 1e-30)) if float(p.grad.abs().max()) > 1e-12 else 0.)
        assert max(errs) < 1e-11, (name, max(errs))
        worst = max(worst, max(errs))
        path = os.path.join(HERE, name + '_fp64.npz')
        np.savez_compressed(path, **fix)
        top = sorted(((v, k) for k, v in sens.items()), reverse=True)[:2]
        print('%-16s loss64 %.12f  pre-activation layers %d  -> %s (%.1f KB); most sensitive to one float32 rounding of the parameters: %s'
              % (name, float(loss), len(pre), os.path.relpath(path, ROOT), os.path.getsize(path) / 1024, [('%.1e' % v, k) for v, k in top]))
    print('oracle(fp64) vs reference(fp64): worst relative difference %.2e' % worst)


if __name__ == '__main__':
    main()
