#!/usr/bin/env python
"""Inference throughput of the hot path: model.eval() under no_grad on resident batches (forward only), and the reference's
evaluation protocol -- evalio.evaluate with 5 test-time passes (train.py:21-91, 83-87) -- over a loader of host-side graphs
(collate + host-to-device copy included).  GPU only.  usage: python tools/eval_bench.py [batch] [graphs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import evalio, network  # noqa: E402
from cgc_net_amd.data import Batch, DataListLoader, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G = int(sys.argv[2]) if len(sys.argv) > 2 else 160
ds = SyntheticCellGraphs(4 * B, 1800, 16, base_seed=0)
batches = [Batch.from_data_list([ds[b * B + i] for i in range(B)]).to(dev) for b in range(4)]
torch.manual_seed(0)
model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True, norm_adj=True,
                                      jk=True, drop_out=0.2).to(dev)
for native in (True, False):
    model.native = native
    model.eval()
    with torch.no_grad():
        for i in range(3):
            model(batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 20
        for i in range(steps):
            model(batches[i % 4])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print('forward only, batch %d, %s: %.3f ms per batch, %.0f graphs/s' % (B, 'sequencer' if native else 'per-operator path', 1e3 * el / steps,
                                                                           B * steps / el))
model.native = True
model.eval()
gen = SyntheticCellGraphs(G, 1800, 16, base_seed=10 ** 6)
items = [gen[i] for i in range(G)]            # (materialised: the generator builds a graph -- k-NN on the host -- per access)
loader = DataListLoader(items, batch_size=B, shuffle=False)
def protocol(passes):
    votes = []
    with torch.no_grad():
        for rep in range(passes):                         # test-time passes (train.py:27-36, 83-87)
            pending = []
            for data in loader:                           # lists of host-side Data: device-side collate (one packed copy + one kernel), as evalio.evaluate does
                pending.append(model(Batch.from_data_list(data, device=dev)))
            votes.append(torch.cat(pending).cpu())        # (evaluate() takes a pass's predictions to the host in ONE transfer: round 6)
    return torch.stack(votes).mean(0).argmax(1)


protocol(1)                                               # warm-up: arenas, pinned staging buffer, kernel modules (round 5 timed these too)
torch.cuda.synchronize()
t0 = time.perf_counter()
pred = protocol(5)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print('evaluation protocol (the loop of evalio.evaluate): %d graphs x 5 test-time passes in %.2f s = %.0f graph-passes/s (host collate + H2D + '
      'D2H of the predictions included; one untimed warm-up pass)' % (G, el, 5 * G / el))

if os.environ.get('EVAL_PROFILE'):                        # where the host spends a pass (cProfile, cumulative)
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    protocol(2)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
