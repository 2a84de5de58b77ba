"""CPU, world_size 2 over gloo: the data-parallel wrapper (shard by cumulative node count, one flat gradient
bucket all-reduced and averaged, parameters broadcast once) against a single-process emulation of the same split."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import cgc_net_amd  # noqa: F401
    import cgc_net_amd.kernels as kernels
    from cgc_net_amd import network
    from cgc_net_amd.data import SyntheticCellGraphs
    from cgc_net_amd.parallel import DataParallel
    from oracle.flat_ref import TorchKernels
    kernels._instance = TorchKernels()      # CPU test seam (tests only)
    torch.manual_seed(100 + rank)           # different initial weights per rank: the broadcast must fix that
    model = network.SoftPoolingGcnEncoder(96, 6, 8, 8, True, True, 8, 3, 0.25, [50], load_data_sparse=True,
                                          norm_adj=True, jk=True)
    dp = DataParallel(model)
    ds = SyntheticCellGraphs(6, 30, num_features=6, base_seed=7)
    items = [ds[i] for i in range(6)]
    dp.train()
    _, loss = dp(items)
    torch.mean(loss).backward()
    rec = {'sd': {k: v.clone() for k, v in model.state_dict().items()},
           'grad': {k: p.grad.clone() for k, p in model.named_parameters()},
           'n_local': len(dp.local_chunk(items)), 'loss': loss.detach()}
    # fewer graphs than ranks (the last partial batch of an epoch): rank 1 idles through the step, still joins the all-reduce,
    # and the mean is over the ONE active replica (torch_geometric's scatter uses fewer devices)
    model.zero_grad()
    _, loss1 = dp(items[:1])
    torch.mean(loss1).backward()
    rec['grad_single'] = {k: p.grad.clone() for k, p in model.named_parameters()}
    rec['sd_single'] = {k: v.clone() for k, v in model.state_dict().items()}
    # evaluation: every rank scores its chunk, results are gathered, every rank reports the metrics of ALL patches
    from cgc_net_amd.data import DataListLoader
    from cgc_net_amd.evalio import ImageLevelVote, evaluate
    ds.idxlist = ['fold/img%d_grade_%d_p%d.pt' % (i // 2, i // 2 % 3 + 1, i) for i in range(6)]
    vote = ImageLevelVote(['img%d_grade_%d' % (i, i % 3 + 1) for i in range(3)])
    rec['eval'] = evaluate(DataListLoader(ds, batch_size=4), dp, vote, test_time=2)
    rec['votes'] = {k: list(v) for k, v in vote.prediction.items()}
    torch.save(rec, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(2))
    for k in r0['grad']:
        assert torch.equal(r0['grad'][k], r1['grad'][k]), k           # identical averaged gradients on every rank
    # parameters were broadcast from rank 0 (BN buffers then evolve per rank, as per replica in the reference)
    for k in r0['sd']:
        if 'running_' not in k and 'num_batches' not in k:
            assert torch.equal(r0['sd'][k], r1['sd'][k]), k
    assert r0['n_local'] + r1['n_local'] == 6
    for k in r0['grad_single']:
        assert torch.equal(r0['grad_single'][k], r1['grad_single'][k]), k
    assert any(float(v.abs().max()) > 0 for v in r0['grad_single'].values())
    assert r0['eval'] == r1['eval'] and r0['votes'] == r1['votes']
    assert sum(len(v) for v in r0['votes'].values()) == 12              # 6 patches x 2 test-time passes, on EVERY rank

    # single-process emulation of the same split with rank 0's weights
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
    import cgc_net_amd  # noqa: F401
    import cgc_net_amd.kernels as kernels
    from cgc_net_amd import network
    from cgc_net_amd.data import Batch, SyntheticCellGraphs, partition_by_nodes
    from oracle.flat_ref import TorchKernels
    old = kernels._instance
    kernels._instance = TorchKernels()
    try:
        ds = SyntheticCellGraphs(6, 30, num_features=6, base_seed=7)
        chunks = partition_by_nodes([ds[i] for i in range(6)], 2)
        grads = []
        for c in chunks:
            m = network.SoftPoolingGcnEncoder(96, 6, 8, 8, True, True, 8, 3, 0.25, [50], load_data_sparse=True,
                                              norm_adj=True, jk=True)
            sd = {k: v for k, v in r0['sd'].items()}
            m.load_state_dict(sd)
            # undo the one BN buffer update rank 0 made: irrelevant for train-mode gradients
            m.train()
            _, loss = m(Batch.from_data_list(c))
            loss.backward()
            grads.append({k: p.grad for k, p in m.named_parameters()})
        for k in r0['grad']:
            want = (grads[0][k] + grads[1][k]) / 2
            assert torch.allclose(r0['grad'][k], want, rtol=1e-5, atol=1e-7), k
        # the single-graph step: gradient of the one active replica, undivided
        m = network.SoftPoolingGcnEncoder(96, 6, 8, 8, True, True, 8, 3, 0.25, [50], load_data_sparse=True, norm_adj=True, jk=True)
        m.load_state_dict(r0['sd'])
        m.train()
        _, loss = m(Batch.from_data_list([ds[0]]))
        loss.backward()
        for k, p in m.named_parameters():
            assert torch.allclose(r0['grad_single'][k], p.grad, rtol=1e-5, atol=1e-7), k
    finally:
        kernels._instance = old
