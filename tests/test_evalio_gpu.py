"""GPU: F4 on the device -- the evaluation protocol (train.py:21-91) and the checkpoint interchange (train.py:202-207,
common/utils.py:82-94) with the model, the forward passes and the optimizer state living on the MI355X; plus the
data-parallel wrapper on a 1-rank RCCL ('nccl') process group (the engine-callback all-reduce runs once on hardware)."""
import os

import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import evalio, kernels, network
from cgc_net_amd.data import Batch, DataListLoader, SyntheticCellGraphs
from oracle import dense_ref
from util import build_model, load_case, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class _WithIdx(torch.utils.data.Dataset):
    def __init__(self, ds, names):
        self.ds, self.idxlist, self.epoch = ds, names, 0

    def __len__(self):
        return len(self.idxlist)

    def __getitem__(self, i):
        d = self.ds[i]
        d.patch_idx = torch.tensor([i])
        return d

    def set_val_epoch(self, e):
        self.epoch = e


def test_evaluate_on_device_matches_oracle_protocol():
    ds = SyntheticCellGraphs(8, 120, 16, base_seed=21)
    names = ['/fold/img%d_grade_%d_patch%d.pt' % (i // 2, 1 + (i // 2) % 3, i) for i in range(8)]
    loader = DataListLoader(_WithIdx(ds, names), batch_size=3)
    gt = ['img%d_grade_%d' % (i, 1 + i % 3) for i in range(4)]
    args = (240, 16, 20, 20, True, True, 20, 3, 0.1, [50])
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.2)
    torch.manual_seed(4)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw)
    model = network.SoftPoolingGcnEncoder(*args, **kw)
    model.load_state_dict(ref.state_dict())
    model.to(DEV).train()
    vote = evalio.ImageLevelVote(gt)
    res = evalio.evaluate(loader, model, vote, test_time=2)
    assert kernels.is_native() and model.training and loader.dataset.epoch == 1
    # the same protocol through the oracle on the host: identical votes and metrics
    ref.eval()
    vote_ref = evalio.ImageLevelVote(gt)
    preds = []
    with torch.no_grad():
        for rep in range(2):
            for lo in range(0, 8, 3):
                items = [ds[i] for i in range(lo, min(lo + 3, 8))]
                out = ref(Batch.from_data_list(items))
                vote_ref.batch_patch_result(names[lo:lo + len(items)], out.argmax(1).numpy())
                preds.append(out)
    assert {k: sorted(v) for k, v in vote.prediction.items()} == {k: sorted(v) for k, v in vote_ref.prediction.items()}
    img_acc, binary_acc = vote_ref.final_result()
    assert res['img_acc'] == img_acc and res['binary_acc'] == binary_acc
    model.eval()
    with torch.no_grad():
        got = torch.cat([model(Batch.from_data_list([ds[i] for i in range(lo, min(lo + 3, 8))]).to(DEV)) for lo in range(0, 8, 3)])
    assert rel_err(got, torch.cat(preds[:3])) < 1e-4


def test_checkpoint_roundtrip_on_device(tmp_path):
    cfg, batch, sd, out, grad, sd3 = load_case('medium_shipped', DEV)
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    evalio.load_reference_state(model, {'state_dict': {'module.' + k: v for k, v in sd.items()}})   # reference-written, DataParallel-prefixed
    model.to(DEV).train()
    _, loss = model(batch)                 # the fixture's protocol (tests/golden/make_golden.py): one plain forward/backward first,
    loss.backward()                        # then three Adam steps -- the BatchNorm buffers see four training forwards
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(2):
        _, loss = model(batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
    f = os.path.join(str(tmp_path), 'run', 'weight.pth.tar')
    evalio.save_checkpoint(evalio.checkpoint_state(model, opt, epoch=1, loss=float(loss.detach()), val_acc=0.5), True, f)
    ck = evalio.load_checkpoint(os.path.join(str(tmp_path), 'run', 'model_best.pth.tar'))
    assert ck['epoch'] == 2 and all(v.device.type == 'cpu' for v in ck['state_dict'].values())
    # resume on the device: a fresh model + optimizer continue to EXACTLY the same third step as the uninterrupted run
    resumed = build_model(network.SoftPoolingGcnEncoder, cfg)
    evalio.load_reference_state(resumed, ck)
    resumed.to(DEV).train()
    opt2 = torch.optim.Adam(resumed.parameters(), lr=1e-3, weight_decay=1e-4)
    opt2.load_state_dict(ck['optimizer'])
    for m, o in ((model, opt), (resumed, opt2)):
        _, loss = m(batch)
        o.zero_grad()
        loss.backward()
        o.step()
    for (k, a), (_, b) in zip(sorted(model.state_dict().items()), sorted(resumed.state_dict().items())):
        assert torch.equal(a, b), k
    # and the uninterrupted run is the reference's trajectory (fixture: state after 3 Adam steps)
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            assert rel_err(v, sd3[k]) < 2e-3, k


def test_data_parallel_wrapper_on_one_rank_rccl_group():
    """World size 1 over 'nccl' (= RCCL): process-group init on the GPU, parameter broadcast, and the flat-bucket
    all-reduce queued by the autograd-engine callback execute on hardware; with one rank the averaged gradients must
    equal the plain module's."""
    import torch.distributed as dist
    from cgc_net_amd.parallel import DataParallel
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + os.getpid() % 1000))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        ds = SyntheticCellGraphs(5, 150, 16, base_seed=31)
        items = [ds[i] for i in range(5)]
        args = (300, 16, 20, 20, True, True, 20, 3, 0.1, [50])
        kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.)
        torch.manual_seed(2)
        plain = network.SoftPoolingGcnEncoder(*args, **kw).to(DEV)
        wrapped_net = network.SoftPoolingGcnEncoder(*args, **kw).to(DEV)
        wrapped_net.load_state_dict(plain.state_dict())
        dp = DataParallel(wrapped_net)
        assert dp.world == 1 and dist.get_backend() == 'nccl'
        dp.world = 2                       # force the multi-rank code path (hooks + all-reduce) on the single rank ...
        dp._active = 2
        for p in dp._params:
            p.register_post_accumulate_grad_hook(dp._on_grad)
        dp.train(), plain.train()
        _, loss = dp(Batch.from_data_list(items).to(DEV))
        torch.mean(loss).backward()
        _, loss_p = plain(Batch.from_data_list(items).to(DEV))
        loss_p.backward()
        torch.cuda.synchronize()
        gp = dict(plain.named_parameters())
        for k, p in wrapped_net.named_parameters():          # ... where SUM over one rank / 2 = half the plain gradient
            assert torch.allclose(p.grad * 2.0, gp[k].grad, rtol=1e-5, atol=1e-8), k
        # the sequencer left every gradient in one buffer: that buffer was all-reduced in place
        assert dp._seq_total is not None and dp._step_buffer() is not None and dp._step_buffer().numel() == dp._seq_total
        # the same collective when the gradients are NOT there (a pass on the per-operator path, an idle rank): gathered into a bucket
        # of the same layout, reduced, scattered back -- halved once more here
        wrapped_net._step_flat = None
        dp._allreduce_grads()
        torch.cuda.synchronize()
        for k, p in wrapped_net.named_parameters():
            assert torch.allclose(p.grad * 4.0, gp[k].grad, rtol=1e-5, atol=1e-8), k
        # a rank that received no graphs: zero loss through every parameter, zero gradients after the (bucket) all-reduce
        for p in wrapped_net.parameters():
            p.grad = None
        _, idle_loss = dp._idle_step()
        idle_loss.backward()
        torch.cuda.synchronize()
        assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in wrapped_net.parameters())
        t = torch.ones(4, device=DEV)
        dist.all_reduce(t)
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


def test_bench_self_launch_two_ranks_reports_strong_and_weak():
    """`python bench.py --gpus 2` (no torchrun) starts its own ranks.  The default N>1 line says what it measures, both ways: the
    headline ``value`` is STRONG scaling (ONE global batch split by cumulative node count, the reference's DataParallel scatter,
    train.py:178-179,276-287) and the same line carries the weak-scaling leg, the HIP-event-timed gradient exchange, the size of the
    process group and the per-rank clocks.  Two ranks share the one GPU of the test box over gloo (RCCL refuses duplicate
    devices); the 8-GPU RCCL run is the driver's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--oversubscribe',
           '--steps', '3', '--warmup', '1', '--batch', '8', '--nodes', '300', '--maxn', '600', '--no-cpu-baseline', '--pool', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                                     # ONE JSON line
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['scaling'] == 'strong' and rec['config']['global_batch'] == 8
    assert rec['value'] > 0 and abs(rec['value'] - 8 * 3 / (rec['ms_per_step'] * 3e-3)) < 0.05 * rec['value']
    assert 'STRONG' in rec['scaling_note']
    assert rec['rccl_ranks'] == 2 and rec['backend'].startswith('gloo') and 'rccl_version' in rec
    # every rank says what it worked on and where it ran: the cumulative-node-count split of each global batch, ONE intra-op thread,
    # a core slice of its own (disjoint from the other rank's when the box has more than one usable core)
    assert [r_['rank'] for r_ in rec['ranks']] == [0, 1]
    assert all(r_['threads'] == 1 and 'cores' in r_ for r_ in rec['ranks'])
    assert all(a + b == 8 for a, b in zip(rec['ranks'][0]['graphs'], rec['ranks'][1]['graphs']))
    assert rec['ranks'][0]['cores'] != rec['ranks'][1]['cores'] or len(os.sched_getaffinity(0)) == 1
    # the weak leg: 8 graphs per rank per step
    assert rec['weak_global_batch'] == 16
    assert rec['weak_value'] > 0 and abs(rec['weak_value'] - 16 * 3 / (rec['weak_ms_per_step'] * 3e-3)) < 0.05 * rec['weak_value']
    # one gradient exchange per step, timed with events, shorter than the step it is part of
    assert rec['allreduce']['per_step'] == 1.0 and rec['allreduce']['bytes'] > 0
    assert 0.0 < rec['allreduce_ms'] < rec['ms_per_step'] and 0.0 < rec['weak_allreduce_ms'] < rec['weak_ms_per_step']
    r = rec['ms_per_step_ranks']
    assert 0.0 < r['min'] <= r['max'] <= rec['ms_per_step'] * 1.001


def test_bench_single_leg_flags():
    """--scaling strong / --scaling weak run one leg only (what profiles/ uses to look at one of them)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--oversubscribe', '--scaling', 'weak',
           '--steps', '2', '--warmup', '1', '--batch', '4', '--nodes', '200', '--maxn', '400', '--no-cpu-baseline', '--pool', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert rec['scaling'] == 'weak' and rec['config']['global_batch'] == 8 and 'strong_value' not in rec and 'weak_value' not in rec


def test_one_epoch_example_runs_the_reference_training_loop():
    """BASELINE.json configs[0]/[1] plumbing: examples/train_synthetic.py = the reference's loop (train.py:174-184: model(data) ->
    torch.mean(cls_loss) -> zero_grad / backward / step, StepLR, evaluate) over 200 synthetic graphs of ~300 nodes, shipped flags
    (parallel_train.sh:2-3: batch 4, --jk --norm_adj --drop 0.2), one epoch, through DataListLoader + DataParallel on the device."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'train_synthetic.py')], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    m = re.search(r'epoch 0: avg loss ([0-9.]+), val acc ([0-9.]+), (\d+) graphs', out.stdout)
    assert m, out.stdout[-2000:]
    loss, acc, seen = float(m.group(1)), float(m.group(2)), int(m.group(3))
    assert seen == 200 and 0.0 < loss < 2.0 and 0.0 <= acc <= 1.0          # (labels are random: the loss sits near ln 3)
