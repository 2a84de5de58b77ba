#!/usr/bin/env python
"""Benchmark of the CGC-Net hot path on MI355X: cell-graphs/sec, forward + backward (+ Adam step), batch 32.

Contract: ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.  N>1 runs one rank per GPU over
RCCL: either the driver starts the ranks (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``) or a plain
``python bench.py --gpus N`` starts them itself.

What an N>1 line measures (``--scaling both``, the default): the metric says "batch=32", and the reference's data parallelism
is torch_geometric's DataParallel -- ONE list of ``batch_size`` graphs scattered over the GPUs by cumulative node count
(train.py:178-179, 276-287) -- so the headline ``value`` is STRONG scaling: a global batch of 32 per step, 32/N graphs per GPU
(4 at N=8), W warm-up + exactly K timed steps.  The same line also carries ``weak_value`` (every GPU processes its own 32
graphs per step: a second leg of W + K steps, reported beside the headline, never instead of it), ``allreduce_ms`` (HIP events
around the gradient exchange of the timed strong steps; mean over steps, max over ranks), ``rccl_ranks`` (the process group's
size and backend as torch.distributed reports them) and ``ms_per_step_ranks`` (min / max over ranks of each rank's own clock).
``--scaling strong`` / ``--scaling weak`` run one leg only.  A "step" is one pass of the hot path over one batch of synthetic
cell graphs that is already resident in HBM: CSR build from ``edge_index`` (the reference densifies here), the full
SoftPoolingGcnEncoder forward, cross-entropy, backward through every kernel, gradient all-reduce (N>1), Adam.

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "C3"): 32 graphs per GPU, ~1800 nodes / ~16k edges / 16 features
each, cluster counts fixed by the reference's ``setting.max_num_nodes`` = 11404 -> C1 = 1140, C2 = 114
(setting.py:15, train.py:254), model flags as shipped in parallel_train.sh (--jk --norm_adj --drop 0.2; ``--flags plain``
turns the three off).  ``--maxn 1800`` gives the "clusters proportional to the graph" variant (C1 = 180).

Extra objects on the line:
  roofline            dominant kernel = the 128x128 pipelined fp32-MFMA GEMM k_gemm_f32<2,2,2,2,*> (bound "mfma", peak
                      157.3 TFLOP/s): algorithmic flops 2MNK of every launch of it / HIP-event duration of that launch,
                      measured inside the timed region (rocprofv3 --stats: average over its three instantiations)
  roofline_aggregation  the wide neighbour-aggregation SpMM A*S ("K4", bound "hbm", peak 8000 GB/s): algorithmic
                      bytes 4(n+1) + 4 nnz (+4 nnz weighted) + 8 n W per launch / HIP-event duration
  cpu_baseline        the dense CPU oracle (the reference's algorithm incl. densification; oracle/dense_ref.py)
                      timed on the host cores, rank 0, N=1 only, on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (matrix)
HBM_PEAK_GBS = 8000.0            # HBM3E spec peak (6.29 TB/s measured float4 copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--batch', type=int, default=32, help="graphs per GPU per step (weak scaling) / per global step (strong scaling)")
    p.add_argument('--scaling', choices=['both', 'weak', 'strong'], default='both',
                   help="N>1 only.  'strong': ONE global batch of --batch graphs per step is split over the GPUs by cumulative node "
                        "count, as the reference's DataParallel scatter does (train.py:276-287); 'weak': every GPU gets its own --batch "
                        "graphs per step; 'both' (default): the strong leg is the headline value, the weak leg is reported beside it "
                        "(weak_value) in the same line")
    p.add_argument('--nodes', type=int, default=1800, help='mean nodes per graph')
    p.add_argument('--feat', type=int, default=16)
    p.add_argument('--maxn', type=int, default=11404, help="the reference's setting.max_num_nodes (fixes cluster counts)")
    p.add_argument('--flags', choices=['plain', 'shipped'], default='shipped',
                   help="'shipped' (default) = the reference's only shipped hyper-parameter set, parallel_train.sh:2-3: "
                        "--jk --norm_adj --drop 0.2; 'plain' = none of the three (SURVEY 8(d) parity configuration)")
    p.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1: 'nccl' (= RCCL over xGMI; default) or 'gloo' (tests)")
    p.add_argument('--oversubscribe', action='store_true',
                   help='tests only: ranks share the visible GPUs (rank %% device_count); needs --backend gloo (RCCL refuses duplicates)')
    p.add_argument('--spatial', action='store_true',
                   help='list the nuclei of every graph grid cell by grid cell (data.spatial_order) instead of in draw order')
    p.add_argument('--pool', type=int, default=4, help='distinct resident batches cycled through')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-warmup', type=int, default=3, help='CPU baseline: untimed warm-up steps')
    p.add_argument('--cpu-steps', type=int, default=5, help='CPU baseline: timed steps')
    p.add_argument('--no-kernel-timing', action='store_true', help='do not record per-launch HIP events')
    p.add_argument('--no-split-leg', action='store_true',
                   help='N=1: skip the second leg that runs the same W + K steps with the six dominant products in mode CGC_GEMM_SPLIT_BF16 '
                        '(reported beside the headline as value_split / roofline_split; the headline is always the exact fp32 kernel)')
    p.add_argument('--gemm-mode', type=int, default=0, help='experiments: mode of the HEADLINE leg (0 exact, 1 split bf16, 2 three fp16 pairs); the JSON says so')
    p.add_argument('--no-gc', action='store_true',
                   help="experiment: Python's cyclic garbage collector off during the timed steps (gc.disable() after a gc.collect()): does "
                        "the idle gap at the forward -> backward turn of some traced steps come from a collection pause?")
    p.add_argument('--plain-adam', action='store_true', help='torch.optim.Adam without fused=True (one kernel per parameter group)')
    return p.parse_args()


def source_hash():
    """sha256 over the kernel sources (cgc-net_amd/csrc/*.hip|*.hpp|Makefile, include/*.h): what the committed counter summaries
    under profiles/ are stamped with, so that a number measured on other code is never quoted for this one."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'cgc-net_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'cgc-net_amd', 'csrc', '*.hpp'))
                   + glob.glob(os.path.join(ROOT, 'include', '*.h')) + [os.path.join(ROOT, 'cgc-net_amd', 'csrc', 'Makefile')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def make_model(args, module):
    kw = dict(concat=True, gcn_name='SAGE', load_data_sparse=True)
    if args.flags == 'shipped':
        kw.update(norm_adj=True, jk=True, drop_out=0.2)
    return module.SoftPoolingGcnEncoder(args.maxn, args.feat, 20, 20, True, True, 20, 3, 0.1, [50], **kw)


def usable_cores():
    """Host cores this process may actually use: min(online CPUs, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def bind_rank(rank, local, world, dev_index):
    """N > 1: one intra-op thread per rank and a core slice of its own -- eight ranks inheriting torch's default intra-op pool would put
    8 x (all cores) threads on the box, and the step is issued by ONE host thread per rank (train.py:276-287 runs one Python thread per
    replica).  The slice comes from the cores this process may use; cores local to the GPU's NUMA node (sysfs local_cpulist of its PCI
    function) are preferred, and ranks whose candidate sets coincide share them out in rank order.  Returns a description for the JSON
    line.  Best effort: any failure leaves the affinity alone and says so."""
    torch.set_num_threads(1)
    info = {'threads': 1}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return dict(info, cores='affinity not supported')
    cand, numa = allowed, None
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
        base = '/sys/bus/pci/devices/' + bdf
        numa = int(open(base + '/numa_node').read())
        loc = set()
        for part in open(base + '/local_cpulist').read().strip().split(','):
            if part:
                lo, _, hi = part.partition('-')
                loc.update(range(int(lo), int(hi or lo) + 1))
        if loc & set(allowed):
            cand = sorted(loc & set(allowed))
    except (OSError, ValueError, AttributeError, RuntimeError):
        pass
    sets = [None] * world
    dist.all_gather_object(sets, (tuple(cand), local))
    peers = sorted(l for c, l in sets if c == tuple(cand))           # the ranks that would pick from the same cores
    per = max(1, len(cand) // len(peers))
    me = peers.index(local)
    mine = cand[me * per:(me + 1) * per] or cand[-1:]
    try:
        os.sched_setaffinity(0, mine)
        info['cores'] = '%d-%d' % (mine[0], mine[-1]) if len(mine) > 1 else str(mine[0])
    except OSError as e:
        info['cores'] = 'unchanged (%s)' % e
    info['numa_node'] = numa
    return info


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args, batches):
    """The dense oracle = the reference's algorithm on the host: densify -> dense convs -> DiffPool -> CE,
    fwd + bwd + Adam, same seeded graphs, all host cores."""
    from oracle import dense_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = make_model(args, dense_ref)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)

    def step(b):
        _, loss = model(b)
        opt.zero_grad()
        loss.mean().backward()
        opt.step()
    n_warm, n_timed = args.cpu_warmup, args.cpu_steps      # >= 3 warm-up + >= 5 timed steps at batch 32
    tw = time.perf_counter()
    for i in range(n_warm):                # warm-up (allocator, thread pool, page faults of the 592 MB dense adjacency)
        step(batches[i % len(batches)])
    tw = time.perf_counter() - tw
    t0 = time.perf_counter()
    for i in range(n_timed):
        step(batches[(n_warm + i) % len(batches)])
    el = time.perf_counter() - t0
    return {'value': round(args.batch * n_timed / el, 3), 'unit': 'graphs/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'cpu_model': cpu_model(),
            'host': '%d logical CPUs online, %d usable (affinity/cgroup quota)' % (os.cpu_count() or 1, cores),
            'sample': '%d timed fwd+bwd+Adam steps of batch %d (%.1f s) after %d warm-up steps (%.1f s), dense '
                      'oracle/dense_ref.py (the reference algorithm incl. densification), same seeded graphs'
                      % (n_timed, args.batch, el, n_warm, tw)}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.gpus > 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: start the ranks ourselves (one process per GPU over RCCL) and pass the line through
        import subprocess
        port = 29500 + os.getpid() % 2000
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.oversubscribe:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    binding = bind_rank(rank, int(os.environ.get('LOCAL_RANK', '0')), world, dev.index) if world > 1 else None      # (the un-wrapped local rank: --oversubscribe folds `local`)

    import cgc_net_amd  # noqa: F401
    from cgc_net_amd import kernels, network
    from cgc_net_amd.data import Batch, SyntheticCellGraphs
    from cgc_net_amd.parallel import DataParallel

    # ---- synthetic workload: `pool` distinct batches, resident in HBM.  weak: per rank, seeded by rank; strong: the SAME
    # global batches on every rank, each rank keeping its chunk of the cumulative-node-count split (data.partition_by_nodes)
    from cgc_net_amd.data import partition_by_nodes
    legs = (['single'] + ([] if args.no_split_leg else ['split', 'half'])) if world == 1 else (['strong', 'weak'] if args.scaling == 'both' else [args.scaling])

    def make_batches(leg):
        strong_ = leg == 'strong'
        ds = SyntheticCellGraphs(args.pool * args.batch, args.nodes, args.feat, base_seed=0 if (strong_ or world == 1) else 100000 * rank,
                                 spatial=args.spatial)
        lists_ = [[ds[b * args.batch + i] for i in range(args.batch)] for b in range(args.pool)]
        if strong_:
            chunks = [partition_by_nodes(l, world) for l in lists_]
            if any(len(c) != world for c in chunks):
                raise SystemExit('cannot split %d graphs over %d ranks' % (args.batch, world))
            lists_ = [c[rank] for c in chunks]
        cpu_ = [Batch.from_data_list(l) for l in lists_]
        return lists_, cpu_, [b.to(dev) for b in cpu_]

    torch.manual_seed(0)
    model = make_model(args, network).to(dev)
    model.train()
    dp = DataParallel(model) if world > 1 else model
    if world > 1:
        dp.time_allreduce(True)
    # the reference's optimiser (common/utils.py:119-121: Adam, lr 1e-3, weight decay 1e-4); fused=True is the same update as
    # ONE kernel over all parameters instead of ~16 multi-tensor launches (host-side cost matters at small per-GPU batches)
    if args.plain_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    else:       # the same update as torch's fused Adam, as ONE launch of the library's cgc_adam_step on the sequencer's flat gradient buffers
        from cgc_net_amd.optim import Adam
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, model=model)
    torch.autograd.set_multithreading_enabled(False)     # backward on the calling thread: no engine-thread hand-off per node

    def step(b):
        _, loss = dp(b)
        if loss.dim():                  # train.py:179 torch.mean(cls_loss): one loss per replica there, a scalar per process here
            loss = torch.mean(loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def run_leg(leg, with_kernel_timing):
        """W untimed + exactly K timed steps, bracketed by barrier + synchronize; returns the leg's measurements."""
        lists_, cpu_, dev_ = make_batches(leg)
        model.gemm_mode = 1 if leg == 'split' else 2 if leg == 'half' else args.gemm_mode      # (cgc_gemm_f32_ws's mode; the headline leg is exact unless asked otherwise)
        for i in range(args.warmup):
            step(dev_[i % len(dev_)])
        kernels.get()
        if world > 1:
            dp.allreduce_ms()               # (drop the warm-up steps' records)
        # HIP events around every launch of the dominant GEMM and of the wide SpMM (the library's measurement hook: recorded on the
        # launch stream, also for launches issued by the step sequencer)
        timer_ = kernels.LaunchTimer(64 * args.steps + 64) if with_kernel_timing else None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if args.no_gc:
            import gc
            gc.collect()
            gc.disable()
        if timer_ is not None:
            timer_.start()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss_ = step(dev_[i % len(dev_)])
        torch.cuda.synchronize()
        own = time.perf_counter() - t0          # this rank's own clock: before it waits for the others
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if timer_ is not None:
            timer_.stop()
        if args.no_gc:
            import gc
            gc.enable()
        res = dict(leg=leg, lists=lists_, cpu_batches=cpu_, timer=timer_, loss=loss_)
        if world > 1:
            ar = dp.allreduce_ms()
            t = torch.tensor([el, own, -own, (sum(ar) / len(ar)) if ar else 0.0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
            res.update(own_max=float(t[1]), own_min=-float(t[2]), allreduce_ms=float(t[3]), allreduce_calls=len(ar))
            info = [None] * world         # what each rank worked on and where it ran: the node-count split, its cores, its NUMA node
            dist.all_gather_object(info, dict(rank=rank, gpu=local, graphs=[len(l) for l in lists_],
                                              nodes=[int(b.x.shape[0]) for b in cpu_], **(binding or {})))
            res['rank_info'] = info
        res['elapsed'] = el
        if not torch.isfinite(loss_).item():
            raise SystemExit('non-finite loss')
        return res

    results = [run_leg(leg, (not args.no_kernel_timing) and (i == 0 or leg in ('split', 'half'))) for i, leg in enumerate(legs)]
    model.gemm_mode = args.gemm_mode
    head = results[0]
    strong = head['leg'] == 'strong'
    elapsed, lists, cpu_batches, timer = head['elapsed'], head['lists'], head['cpu_batches'], head['timer']
    nodes = sum(b.x.shape[0] for b in cpu_batches) / len(cpu_batches)
    edges = sum(b.edge_index.shape[1] for b in cpu_batches) / len(cpu_batches)

    if rank == 0:
        graphs = args.batch * (1 if strong else world) * args.steps
        c1 = int(args.maxn * 0.1)
        out = {
            'metric': 'cell-graphs/sec (fwd+bwd), batch=32, ~1800 nodes/16 feat',
            'value': round(graphs / elapsed, 2), 'unit': 'graphs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('C3' if args.nodes < 4000 else 'C5 (stress)') + ': full CGC-Net (3 conv blocks + 2 DiffPool) fwd+bwd+Adam, %d graphs/GPU/step, '
                                   '~%d nodes, ~%d edges/graph, %d feat, max_num_nodes=%d (C1=%d, C2=%d), flags=%s'
                                   % (len(lists[0]), round(nodes / len(lists[0])), round(edges / len(lists[0])), args.feat,
                                      args.maxn, c1, int(c1 * 0.1), args.flags),
                       'global_batch': args.batch * (1 if strong else world), 'nodes_per_batch': round(nodes),
                       'parallelism': 'dp%d' % world, 'includes': 'CSR build + fwd + loss + bwd + grad all-reduce + Adam',
                       'step_sequencer': bool(getattr(model, 'native', False)), 'node_order': 'grid cells' if args.spatial else 'draw order', 'fused_adam': not args.plain_adam,
                       'optimiser': 'torch.optim.Adam' if args.plain_adam else 'cgc_adam_step (one launch; torch fused Adam arithmetic)'},
        }
        if world > 1:
            out['scaling_note'] = ("value = STRONG scaling: one global batch of %d graphs per step split over the %d ranks by cumulative node "
                                   "count (the reference's DataParallel scatter, train.py:276-287)" % (args.batch, world)) if strong else \
                                  'value = WEAK scaling: %d graphs per GPU per step' % args.batch
            out['rccl_ranks'] = dist.get_world_size()
            try:
                out['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                                  # (gloo-only builds)
                out['rccl_version'] = None
            out['ranks'] = head['rank_info']
            out['backend'] = dist.get_backend() + (' (= RCCL over xGMI on ROCm)' if dist.get_backend() == 'nccl' else '')
            out['allreduce_ms'] = round(head['allreduce_ms'], 4)
            out['allreduce'] = {'per_step': head['allreduce_calls'] / max(args.steps, 1), 'bytes': 4 * (getattr(dp, '_seq_total', None) or sum(p.numel() for p in model.parameters())),
                                'op': 'AVG in place on the step sequencer\'s flat gradient buffer' if getattr(dp, '_has_avg', False) else 'SUM + divide',
                                'timing': 'HIP events on the launch stream around dist.all_reduce, mean over the timed steps, max over ranks'}
            out['ms_per_step_ranks'] = {'min': round(1e3 * head['own_min'] / args.steps, 3), 'max': round(1e3 * head['own_max'] / args.steps, 3)}
            for r in results[1:]:
                k = r['leg']
                g_ = args.batch * (1 if k == 'strong' else world) * args.steps
                out[k + '_value'] = round(g_ / r['elapsed'], 2)
                out[k + '_ms_per_step'] = round(1e3 * r['elapsed'] / args.steps, 3)
                out[k + '_allreduce_ms'] = round(r['allreduce_ms'], 4)
                out[k + '_global_batch'] = args.batch * (1 if k == 'strong' else world)
        shipped = args.flags == 'shipped'

        def rooflines(timer_):
            """(dominant GEMM, wide SpMM) roofline objects from a leg's per-launch HIP events."""
            recs = timer_.records()
            per_step = len(recs) // args.steps if args.steps and len(recs) % max(args.steps, 1) == 0 else 0
            g_ms = g_fl = s_ms = s_by = 0.0
            g_n = s_n = 0
            for i, (tag, dims, ms) in enumerate(recs):
                if not per_step:
                    break
                bi = (i // per_step) % len(cpu_batches)             # the batch this launch worked on
                nb_nodes, nb_edges = cpu_batches[bi].x.shape[0], cpu_batches[bi].edge_index.shape[1]
                if tag == 'gemm_128x128':
                    g_ms += ms
                    g_fl += kernels.LaunchTimer.gemm_flops(dims, nb_nodes)
                    g_n += 1
                elif tag == 'spmm_wide':
                    n_, width = dims[0], dims[1]
                    # algorithmic bytes: 8 n W (read X once, write Y once) + 4(n+1) + 4 nnz (+ 4 nnz weights; the re-normalised
                    # adjacency also holds every diagonal entry)
                    nnz = nb_edges + (nb_nodes if shipped else 0)
                    s_ms += ms
                    s_by += 8.0 * n_ * width + 4.0 * (n_ + 1) + (8.0 if dims[3] else 4.0) * nnz
                    s_n += 1
            gemm = spmm = None
            if g_n:
                tf = g_fl / (g_ms * 1e-3) / 1e12
                gemm = {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None,
                        'launches_per_step': g_n / args.steps, 'avg_launch_ms': round(g_ms / g_n, 4),
                        'gflop_per_launch': round(g_fl / g_n / 1e9, 3), 'ms_per_step': round(g_ms / args.steps, 3)}
            if s_n:
                gbs = s_by / (s_ms * 1e-3) / 1e9
                spmm = {'kernel': 'k_spmm_wide<9,*> (A*S and its transpose, width %d)' % c1, 'bound': 'hbm',
                        'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None,
                        'launches_per_step': s_n / args.steps, 'avg_launch_ms': round(s_ms / s_n, 4), 'mb_per_launch': round(s_by / s_n / 1e6, 2)}
            timer_.close()
            return gemm, spmm
        EXACT_KERNEL = ('k_gemm_f32<2,2,2,2,*> (fp32 MFMA 32x32x2, 128x128x32 tile; all launches of its NN/NT/TN instantiations, '
                        'tail fix-up kernel included)')
        SPLIT_KERNEL = ('k_gemm_split<*> (the same six products per step as six v_mfma_f32_32x32x16_bf16 pairs per fp32 product, 256x128x16 tile; '
                        'tail fix-up kernel included)')
        if timer is not None:
            gemm, spmm = rooflines(timer)
            if gemm is not None:
                out['roofline'] = dict({'kernel': SPLIT_KERNEL if args.gemm_mode == 1 else EXACT_KERNEL}, **gemm)
            if spmm is not None:
                out['roofline_aggregation'] = spmm
        if args.gemm_mode:
            out['config']['gemm_mode'] = '%s (experiment: the headline leg itself ran in this mode)' % {1: 'CGC_GEMM_SPLIT_BF16', 2: 'CGC_GEMM_SPLIT_F16'}[args.gemm_mode]
        for r in results[1:]:
            if r['leg'] != 'split':
                continue
            # the same W + K steps with the six dominant products in mode CGC_GEMM_SPLIT_BF16: reported BESIDE the headline
            out['value_split'] = round(args.batch * args.steps / r['elapsed'], 2)
            out['ms_per_step_split'] = round(1e3 * r['elapsed'] / args.steps, 3)
            if r['timer'] is not None:
                gemm, _ = rooflines(r['timer'])
                if gemm is not None:
                    tf = gemm['achieved']
                    # the roofline of THIS kernel is the bf16 matrix pipe: it issues six bf16 products per fp32 product
                    out['roofline_split'] = dict({'kernel': SPLIT_KERNEL}, **gemm)
                    out['roofline_split'].update({
                        'achieved': round(6.0 * tf, 1), 'peak': 2500.0, 'frac': round(6.0 * tf / 2500.0, 4),
                        'achieved_is': 'bf16 TFLOP/s issued: 6 pairs x 2MNK of the fp32 product / launch duration, against the dense bf16 MFMA peak',
                        'fp32_equivalent_tflops': tf, 'fp32_equivalent_over_fp32_mfma_peak': round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                        'error_table': 'profiles/r06_split_gemm_error_table.txt (max / rms error vs float64 next to the exact kernel, every form)'})
        HALF_KERNEL = ('k_gemm_absmax<*> + k_gemm_half<*> (the same six products per step as three v_mfma_f32_32x32x16_f16 pairs per fp32 product '
                       'of operands scaled per batch item, 256x128x16 tile; the operand-maximum pass and the tail fix-up kernel included)')
        for r in results[1:]:
            if r['leg'] != 'half':
                continue
            # ... and in mode CGC_GEMM_SPLIT_F16
            out['value_half'] = round(args.batch * args.steps / r['elapsed'], 2)
            out['ms_per_step_half'] = round(1e3 * r['elapsed'] / args.steps, 3)
            if r['timer'] is not None:
                gemm, _ = rooflines(r['timer'])
                if gemm is not None:
                    tf = gemm['achieved']
                    out['roofline_half'] = dict({'kernel': HALF_KERNEL}, **gemm)
                    out['roofline_half'].update({
                        'achieved': round(3.0 * tf, 1), 'peak': 2500.0, 'frac': round(3.0 * tf / 2500.0, 4),
                        'achieved_is': 'fp16 TFLOP/s issued: 3 pairs x 2MNK of the fp32 product / duration of the product (maximum pass + kernel + fix-up), against the dense 16-bit MFMA peak',
                        'fp32_equivalent_tflops': tf, 'fp32_equivalent_over_fp32_mfma_peak': round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                        'error_table': 'profiles/r06_half_gemm_error_table.txt (max / rms error vs float64 next to the exact kernel, every form)'})
        # HBM traffic per launch and counter-derived matrix-core utilisation: from the committed PMC passes of this same command
        # (profiles/make_traffic_json.py, profiles/make_counters_json.py) -- ONLY when they were taken on exactly this kernel source
        # (sha256 over csrc/ + include/, stamped into the json); otherwise null + "stale"
        if args.flags == 'shipped' and args.maxn == 11404 and args.batch == 32 and args.nodes == 1800:
            src = source_hash()
            import glob
            for key in ('traffic', 'counters'):
                found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s.json' % key)))      # the latest round's
                if not found:
                    continue
                fname = os.path.basename(found[-1])
                try:
                    js = json.load(open(found[-1]))
                except (OSError, ValueError):
                    continue
                fresh = js.get('source_sha256') == src
                for obj, tag in (('roofline', 'gemm_split' if args.gemm_mode == 1 else 'gemm_128x128'), ('roofline_aggregation', 'spmm_wide'),
                                 ('roofline_split', 'gemm_split'), ('roofline_half', 'gemm_half')):
                    if obj not in out or tag not in js:
                        continue
                    if key == 'traffic':
                        if fresh:
                            out[obj]['traffic'] = round(js[tag]['hbm_bytes_per_launch'] + (js.get('gemm_half_absmax', {}).get('hbm_bytes_per_launch', 0.0) if tag == 'gemm_half' else 0.0))
                            out[obj]['traffic_source'] = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch; same kernel source)' % fname
                        else:
                            out[obj]['traffic_stale'] = True
                    elif tag in ('gemm_128x128', 'gemm_split', 'gemm_half') and js[tag].get('mfma_busy') is not None:
                        if fresh:
                            out[obj]['mfma_busy_counter'] = js[tag]['mfma_busy']
                            out[obj]['mfma_busy_source'] = ('profiles/%s: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8) of this '
                                                            'command under rocprofv3 --pmc, same kernel source (clock-independent; frac = '
                                                            'mfma_busy x sustained clock / 2.4 GHz)' % fname)
                        else:
                            out[obj]['mfma_busy_stale'] = True
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, cpu_batches)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
