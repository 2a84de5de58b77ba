#!/usr/bin/env python
"""How far apart are two fp32 evaluations of one reference fixture, and why?  For a fixture (default medium_plain) with every product
forced onto the 128 x 128 route: the model is run in the three GEMM modes; per pair of modes the number of max-readout winners taken
differently (the discrete decision of model/network.py:264) and the largest strict distance between the two gradient sets, next to
each mode's distance to the reference's own fp32 gradients (tests/golden/<name>.npz).  GPU only.
usage: python tools/decision_noise.py [fixture ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402

import discrete  # noqa: E402
from cgc_net_amd import kernels, network  # noqa: E402
from util import build_model, load_case  # noqa: E402

DEV = 'cuda:0'
K = kernels.get()
K.lib.cgc_gemm_tuning(11)
NAMES = {0: 'exact', 1: 'bf16 x 6', 2: 'fp16 x 3'}


def strict(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


for name in (sys.argv[1:] or ['medium_plain']):
    cfg, batch, sd, out, grad, _ = load_case(name, DEV)
    runs = {}
    for mode in (0, 1, 2):
        m = build_model(network.SoftPoolingGcnEncoder, cfg, collect_assign=True)
        m.load_state_dict(sd)
        m.to(DEV).train()
        m.native, m.reorder_large, m.gemm_mode = False, False, mode
        with discrete.record_hip_decisions(m) as dec:
            logits, loss = m(batch)
            loss.backward()
            torch.cuda.synchronize()
        runs[mode] = (torch.cat([a.flatten() for _, a in dec.winners]).clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, logits.detach())
    nw = runs[0][0].numel()
    print('%s: %d readout winners per evaluation' % (name, nw))
    for mode in (0, 1, 2):
        w = max((strict(g, grad[k]), k) for k, g in runs[mode][1].items() if float(grad[k].abs().max()) > 1e-9 and not k.endswith('att.bias'))
        print('  %-9s vs the reference\'s fp32 fixture: logits %.1e, worst gradient %.2e (%s)' % (NAMES[mode], strict(runs[mode][2], out['logits']), w[0], w[1]))
    for a, b in ((0, 1), (0, 2), (1, 2)):
        dw = int((runs[a][0] != runs[b][0]).sum())
        w = max((strict(runs[a][1][k], runs[b][1][k]), k) for k in runs[a][1] if float(grad[k].abs().max()) > 1e-9 and not k.endswith('att.bias'))
        print('  %-9s vs %-9s: %d winners taken differently, logits %.1e, worst gradient distance %.2e (%s)' % (NAMES[a], NAMES[b], dw,
              strict(runs[a][2], runs[b][2]), w[0], w[1]))
