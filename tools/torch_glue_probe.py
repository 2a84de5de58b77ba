"""Which Python call sites still make torch launch kernels inside a training step (fills, copies, reductions, the optimiser)?
Runs the bench step under torch.profiler with stacks and prints, per aten op that launched a device kernel, the innermost frames
of this repository.  Usage (GPU box): python tools/torch_glue_probe.py [--batch 4]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cgc_net_amd import network
from cgc_net_amd.data import Batch, SyntheticCellGraphs
from cgc_net_amd.optim import Adam

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--steps', type=int, default=4)
a = ap.parse_args()
sys.argv = [sys.argv[0], '--batch', str(a.batch)]
args = bench.parse()
dev = torch.device('cuda:0')
ds = SyntheticCellGraphs(2 * args.batch, args.nodes, args.feat, base_seed=0)
batches = [Batch.from_data_list([ds[b * args.batch + i] for i in range(args.batch)]).to(dev) for b in range(2)]
torch.manual_seed(0)
model = bench.make_model(args, network).to(dev).train()
opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
torch.autograd.set_multithreading_enabled(False)


def step(b):
    _, loss = model(b)
    loss = torch.mean(loss)
    opt.zero_grad()
    loss.backward()
    opt.step()


for i in range(4):
    step(batches[i % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(a.steps):
        step(batches[i % 2])
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sites = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith('CPU') and ev.name.startswith('aten::') and ev.kernels:
        frames = [f for f in (ev.stack or []) if root in f or 'bench.py' in f or 'torch_glue_probe' in f][:3]
        sites[(ev.name, ' <- '.join(f.replace(root + '/', '') for f in frames) or '(autograd engine / optimiser)',
               ','.join(sorted({k.name[:40] for k in ev.kernels})))] += 1
print('launches per step by aten op and call site (%d steps)' % a.steps)
tot = 0
for (name, where, ks), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print('%6.2f  %-28s %s   [%s]' % (n / a.steps, name, where, ks))
    tot += n
print('total %.1f torch launches per step' % (tot / a.steps))
