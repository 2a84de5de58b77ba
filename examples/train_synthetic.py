#!/usr/bin/env python
"""The reference's training step (train.py:138-244, the loop at :174-184) on synthetic cell graphs, with this package in
place of `model.network` / torch_geometric -- BASELINE.json configs[0]/[1]: 200 graphs, ~300 nodes, 16 features, k-NN edges,
3 classes, one epoch.

    python examples/train_synthetic.py                       # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py   # 8 GPUs (RCCL)

What is kept from the reference: model constructor call (train.py:254-261), DataListLoader protocol (lists of Data),
DataParallel(model)(data) -> (output, cls_loss), torch.mean(cls_loss), Adam(lr 1e-3, wd 1e-4) (common/utils.py:119-121),
StepLR (train.py:146-147), evaluate() accuracy with model.eval() (train.py:21-53).  What is gone: hard-coded .cuda() calls.
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import network  # noqa: E402
from cgc_net_amd.data import DataListLoader, SyntheticCellGraphs  # noqa: E402
from cgc_net_amd.parallel import DataParallel  # noqa: E402


def evaluate(loader, model):
    model.eval()
    correct = total = 0
    with torch.no_grad():
        for data in loader:
            ypred = model(data)
            labels = torch.cat([d.y for d in model.local_chunk(data)]).to(ypred.device)
            correct += int((ypred.argmax(1) == labels).sum())
            total += labels.numel()
    model.train()
    return correct / max(total, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', type=int, default=200)
    ap.add_argument('--nodes', type=int, default=300)
    ap.add_argument('--max-num-nodes', type=int, default=600)      # setting.max_num_nodes: fixes the cluster counts
    ap.add_argument('--batch-size', type=int, default=4)           # parallel_train.sh:2
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--plain', action='store_true', help='without --jk --norm_adj --drop 0.2')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=device)

    train_set = SyntheticCellGraphs(args.graphs, args.nodes, num_features=16, base_seed=0)
    val_set = SyntheticCellGraphs(max(args.graphs // 5, 8), args.nodes, num_features=16, base_seed=10 ** 6)
    # every rank iterates the same global batches; DataParallel keeps this rank's cumulative-node-count chunk
    train_loader = DataListLoader(train_set, batch_size=args.batch_size * world, shuffle=False)
    val_loader = DataListLoader(val_set, batch_size=args.batch_size * world, shuffle=False)

    torch.manual_seed(0)
    flags = dict() if args.plain else dict(norm_adj=True, jk=True, drop_out=0.2)
    model = network.SoftPoolingGcnEncoder(args.max_num_nodes, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True,
                                          gcn_name='SAGE', load_data_sparse=True, **flags)
    model = DataParallel(model.to(device))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=10, gamma=0.1)

    for epoch in range(args.epochs):
        train_loader.dataset.set_epoch(epoch)
        model.train()
        t0, seen, avg = time.time(), 0, 0.0
        for batch_idx, data in enumerate(train_loader):
            _, cls_loss = model(data)
            loss = torch.mean(cls_loss)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            avg += float(loss.detach())
            seen += len(data)
        scheduler.step()
        torch.cuda.synchronize()
        acc = evaluate(val_loader, model)
        if not dist.is_initialized() or dist.get_rank() == 0:
            print('epoch %d: avg loss %.4f, val acc %.3f, %d graphs in %.2f s (%.0f graphs/s incl. host collate + H2D)' % (
                epoch, avg / (batch_idx + 1), acc, seen, time.time() - t0, seen / (time.time() - t0)))
    state = model.module.state_dict()          # train.py:204
    assert 'GCN_pool_1.lin.weight' in state
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
