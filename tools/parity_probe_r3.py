import sys, os, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
import cgc_net_amd
from cgc_net_amd import network
from cgc_net_amd.data import Batch, SyntheticCellGraphs
from oracle import dense_ref
from util import CASES, build_model, load_case, rel_err, elementwise_excess
DEV='cuda:0'
def strict(a,b):
    a,b=a.detach().double().cpu(), b.detach().double().cpu()
    return float((a-b).abs().max()/b.abs().max().clamp_min(1e-30))
for name in CASES:
    cfg, batch, sd, out, grad, sd3 = load_case(name, DEV)
    model = build_model(network.SoftPoolingGcnEncoder, cfg, collect_assign=True)
    model.load_state_dict(sd); model.to(DEV).train()
    logits, loss = model(batch); loss.backward()
    worst = max((strict(p.grad, grad[k]), k) for k,p in model.named_parameters() if float(grad[k].abs().max())>1e-12 and not k.endswith('att.bias'))
    print(name, 'logits excess %.3f' % elementwise_excess(logits, out['logits']), 'assign excess', ['%.3f' % elementwise_excess(s, out['assign%d'%(i+1)]) for i,s in enumerate(model.assign_matrix)], 'worst strict grad %.2e %s' % worst)
for name in ['tiny_shipped','medium_plain','medium_shipped']:
    cfg, batch, sd, out, grad, sd3 = load_case(name, DEV)
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    model.load_state_dict(sd); model.to(DEV).train()
    _, loss = model(batch); loss.backward()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(3):
        _, loss = model(batch); opt.zero_grad(); torch.mean(loss).backward(); opt.step()
    w = max((rel_err(v, sd3[k]), strict(v, sd3[k]), k) for k,v in model.state_dict().items() if v.dtype.is_floating_point)
    model.eval()
    with torch.no_grad(): e = rel_err(model(batch), out['eval_logits3'])
    print(name, '3 adam: worst rel_err %.2e strict %.2e %s; eval logits %.2e' % (w[0], w[1], w[2], e))
# GIN vs fp64
ds = SyntheticCellGraphs(6, 300, num_features=16, base_seed=42)
cpu_batch = Batch.from_data_list([ds[i] for i in range(6)])
args = (600, 16, 20, 20, True, True, 20, 3, 0.1, [50]); kw = dict(concat=True, load_data_sparse=True, drop_out=0., gcn_name='GIN')
torch.manual_seed(3)
ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw)
model = network.SoftPoolingGcnEncoder(*args, **kw); model.load_state_dict(ref.state_dict()); model.to(DEV).train(); ref.train()
ref64 = copy.deepcopy(ref).double()
logits, loss = model(cpu_batch.to(DEV)); loss.backward()
rl, rloss = ref(cpu_batch); rloss.backward()
adj = dense_ref.to_dense_adj(cpu_batch.edge_index, cpu_batch.batch); xd, counts = dense_ref.to_dense_batch(cpu_batch.x, cpu_batch.batch)
ref64.load_data_sparse = False
l64, loss64 = ref64((xd.double(), adj.double(), counts, cpu_batch.y)); loss64.backward()
g32 = dict(ref.named_parameters()); g64 = dict(ref64.named_parameters())
rows = sorted(((strict(p.grad, g64[k].grad), strict(g32[k].grad, g64[k].grad), k) for k,p in model.named_parameters()), reverse=True)
print('GIN vs fp64: hip / fp32-oracle', [( '%.1e'%a, '%.1e'%b, k) for a,b,k in rows[:5]])
print('GIN logits vs fp64 %.2e, fp32 oracle vs fp64 %.2e' % (strict(logits, l64), strict(rl, l64)))
