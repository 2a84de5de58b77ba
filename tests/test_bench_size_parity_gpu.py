"""GPU: oracle parity AT THE SIZES THAT ARE BENCHMARKED (BASELINE.json configs[2] "C3" and configs[4] "C5").

* every contraction shape of the C3 step that ``gemm_dispatch`` (csrc/gemm.hip) sends to the dominant 128x128 kernel
  ``k_gemm_f32<2,2,2,2,*>`` -- NN / NT / TN, flat, ragged-M, ragged-K, padded row strides (ld = 1152), extra K segments,
  accumulate mode, split-K -- against an fp64 product (the kernel the bench line's ``roofline`` is quoted on);
* the FULL model, forward + backward, at C3 shapes (B = 8 graphs of ~1800 nodes, max_num_nodes = 11404 -> C1 = 1140:
  the per-graph contractions launch 9 x 8 x 9 = 648 >= 448 tiles, so all six dominant contractions and the W = 1140 wide
  SpMM are on the path) and at C5 shapes (B = 3 graphs of ~8000 nodes kept by the 'fuse' sampler out of 16000 nuclei, 64 features, max_num_nodes = 16000 ->
  C1 = 1600) against the dense CPU oracle (oracle/dense_ref.py; model/network.py:245-291, parallel_train.sh:2-3).

Tolerances: outputs 1e-4 (max-norm AND elementwise, tests/util.py), gradients 5e-4."""
import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import kernels, network
from cgc_net_amd.data import Batch, Data, SyntheticCellGraphs, radius_graph, sample_nodes_batch
from oracle import dense_ref
import discrete
from util import elementwise_excess, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
C1, LD = 1140, 1152


def rnd(*shape, seed=0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def padded(t, ld=LD):
    """[n, F] rows on the device with a row stride of ``ld`` floats (what ops._wide hands the kernels)."""
    buf = torch.full((t.shape[0], ld), float('nan'), device=DEV)          # NaN padding: any read of it would poison the result
    buf[:, :t.shape[1]] = t.to(DEV)
    return buf[:, :t.shape[1]]


def check(got, want64, what):
    got = got.detach().cpu().double()
    assert torch.isfinite(got).all(), what
    assert rel_err(got, want64) < 2e-5, (what, rel_err(got, want64))
    assert elementwise_excess(got, want64, 1e-4) <= 1.0, (what, elementwise_excess(got, want64, 1e-4))


class _Spy(object):
    """Counts the launches that gemm_dispatch sends to the 128x128 kernel and the wide SpMM launches: the library's measurement
    hook (kernels.LaunchTimer), so launches issued by the step sequencer are seen as well."""

    def __init__(self):
        self.t = kernels.LaunchTimer(512)

    def start(self):
        self.t.start()

    def stop(self):
        self.t.stop()

    @property
    def records(self):
        torch.cuda.synchronize()
        return self.t.counts()


@pytest.fixture
def spy():
    s = _Spy()
    s.start()
    yield s
    s.stop()
    s.t.close()


COUNTS = [1790, 2150, 1475, 1803, 1999, 1644, 2161, 1440]     # B = 8 ragged graphs (C3 node counts): 9 x 8 x 9 = 648 tiles >= 448;
#                                                              a multiple of 8 also takes the K-balanced XCD dealing of gemm_map_tile
GPTR = np.cumsum([0] + COUNTS)
NTOT = int(GPTR[-1])
NB = len(COUNTS)


def test_dominant_gemm_flat_nn_nt_bias_padded_rows(spy):
    K = kernels.get()
    n = 8192
    x, w, bias = rnd(n, C1, seed=1), rnd(C1, C1, seed=2), rnd(C1, seed=3)
    xp = padded(x)
    want = x.double() @ w.double() + bias.double()
    # NN, B = W^T copy on padded rows (ops._LinearCat tall path), bias, output on padded rows
    wp, y = padded(w), padded(torch.zeros(n, C1))
    K.gemm(xp, wp, y, n, C1, C1, False, False, LD, LD, LD, 1.0, 0.0, bias.to(DEV))
    check(y, want, 'NN padded')
    # NT: B stored [N, K] (nn.Linear layout)
    wt = w.t().contiguous().to(DEV)
    y2 = torch.empty(n, C1, device=DEV)
    K.gemm(xp, wt, y2, n, C1, C1, False, True, LD, C1, C1, 1.0, 0.0, bias.to(DEV))
    check(y2, want, 'NT')
    # NT with an extra K segment (Linear over cat[x12 | x3], model/network.py:118-122) and alpha/beta
    x12, w12 = rnd(n, 40, seed=4), rnd(C1, 40, seed=5)
    wcat = torch.cat([w12, w.t()], dim=1).contiguous().to(DEV)            # [out, 40 + 1140]
    c0 = rnd(n, C1, seed=6)
    y3 = c0.clone().to(DEV)
    K.gemm(xp, wcat[:, 40:], y3, n, C1, C1, False, True, LD, 1180, C1, 0.5, -2.0, None,
           extra=[(x12.to(DEV), wcat, 40, 1180, 40, 0, 0)])
    check(y3, 0.5 * (x.double() @ w.double() + x12.double() @ w12.double().t()) - 2.0 * c0.double(), 'NT + extra segment, beta')
    assert spy.records.get('gemm_128x128', 0) == 3


def test_dominant_gemm_ragged_k_tn(spy):
    """S^T P and S^T X per graph (model/network.py:206-207): ragged K = the graph's node count, TN, padded rows."""
    K = kernels.get()
    S, P = rnd(NTOT, C1, seed=1), rnd(NTOT, C1, seed=2)
    gptr = torch.tensor(GPTR, dtype=torch.int32, device=DEV)
    out = torch.full((NB, C1, C1), float('nan'), device=DEV)
    K.gemm(padded(S), padded(P), out, C1, C1, 0, True, False, LD, LD, C1, 1.0, 0.0, None, NB, 0, 0, C1 * C1, gptr, 2,
           max(COUNTS), NTOT)
    want = torch.stack([S[GPTR[b]:GPTR[b + 1]].double().t() @ P[GPTR[b]:GPTR[b + 1]].double() for b in range(NB)])
    check(out, want, 'ragged-K TN')
    assert spy.records.get('gemm_128x128', 0) == 1


def test_dominant_gemm_ragged_m_nn_and_nt_accumulate(spy):
    """dP = S dA' (ragged M, NN) and dS += P dA'^T + X dX'^T (ragged M, NT, beta = 1, extra segment)."""
    K = kernels.get()
    S, P, X = rnd(NTOT, C1, seed=1), rnd(NTOT, C1, seed=2), rnd(NTOT, 60, seed=3)
    dA, dX, Z0 = rnd(NB, C1, C1, seed=4), rnd(NB, C1, 60, seed=5), rnd(NTOT, C1, seed=6)
    gptr = torch.tensor(GPTR, dtype=torch.int32, device=DEV)
    dp = padded(torch.zeros(NTOT, C1))
    K.gemm(padded(S), dA.to(DEV), dp, 0, C1, C1, False, False, LD, C1, LD, 1.0, 0.0, None, NB, 0, C1 * C1, 0, gptr, 1,
           max(COUNTS), NTOT)
    want = torch.cat([S[GPTR[b]:GPTR[b + 1]].double() @ dA[b].double() for b in range(NB)])
    check(dp, want, 'ragged-M NN')
    ds = padded(Z0)
    K.gemm(padded(P), dA.to(DEV), ds, 0, C1, C1, False, True, LD, C1, LD, 1.0, 1.0, None, NB, 0, C1 * C1, 0, gptr, 1,
           max(COUNTS), NTOT, extra=[(X.to(DEV), dX.to(DEV), 60, 60, 60, 0, C1 * 60)])
    want = Z0.double() + torch.cat([P[GPTR[b]:GPTR[b + 1]].double() @ dA[b].double().t() +
                                    X[GPTR[b]:GPTR[b + 1]].double() @ dX[b].double().t() for b in range(NB)])
    check(ds, want, 'ragged-M NT accumulate + extra')
    assert spy.records.get('gemm_128x128', 0) == 2


def test_dominant_gemm_weight_gradient_tn_flat_and_split(spy):
    """dW = dy^T x over all rows (TN, flat) -- direct and through the row-split + deterministic combine of ops.gemm_tn_rows."""
    from cgc_net_amd import ops
    K = kernels.get()
    n = 57600
    dy, x = rnd(n, C1, seed=1), rnd(n, C1, seed=2)
    dyp, xp = padded(dy), padded(x)
    want = dy.double().t() @ x.double()
    out = torch.empty(C1, C1, device=DEV)
    K.gemm(dyp, xp, out, C1, C1, n, True, False, LD, LD, C1)
    check(out, want, 'TN flat')
    out2 = torch.empty(C1, C1, device=DEV)
    ops.gemm_tn_rows(dyp, LD, C1, xp, LD, C1, n, out2)
    check(out2, want, 'TN row-split')
    assert spy.records.get('gemm_128x128', 0) >= 1


def _compare_model(cpu_batch, maxn, feat, flags):
    """tests/discrete.py::compare_model with the kernel-name spy of this file: returns (launch counts, worst gradient)."""
    spy = _Spy()
    worst = discrete.compare_model(cpu_batch, maxn, feat, flags, timer=spy)
    rec = spy.records
    spy.t.close()
    return rec, worst


@pytest.mark.parametrize('flags', [dict(norm_adj=True, jk=True), dict()], ids=['shipped', 'plain'])
def test_full_model_c3_shapes_vs_oracle(flags, gemm_mode):
    ds = SyntheticCellGraphs(8, 1800, 16, base_seed=11)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(8)])
    records, worst = _compare_model(cpu_batch, 11404, 16, flags)
    gemm_mode.check_applied(12)      # (twin + model: six products each)
    # the benchmarked kernels were on the path: the six 128x128 contractions and both wide aggregations
    assert records.get('gemm_128x128', 0) >= 6, records
    assert records.get('spmm_wide', 0) == 2, records


def test_full_model_c5_shapes_fuse_sampled_vs_oracle(gemm_mode):
    """BASELINE configs[4]: ~8000-node graphs (the 'fuse' sampler keeps half of 16000 nuclei, dataflow/data.py:210-219),
    64 features, cluster counts 1600 / 160, shipped flags.  Sampling and the k-NN graph run on the device (F3, F2)."""
    B, cand = 3, 16000
    rng = np.random.RandomState(5)
    side = float(np.sqrt(cand * 1784.0 / 2.0))
    pos = torch.from_numpy(rng.uniform(0.0, side, size=(B * cand, 2)).astype(np.float32)).to(DEV)
    torch.manual_seed(7)
    keep, ks = sample_nodes_batch(pos, [cand] * B, 0.5, 'fuse', generator=None, start=[3, 11, 7])
    assert ks == [8000] * B and keep.numel() == 8000 * B
    items, off = [], 0
    for b in range(B):
        p = pos[keep[off:off + ks[b]]]
        off += ks[b]
        ei = radius_graph(p, 100.0, None, True, 8).cpu()
        x = torch.from_numpy(rng.standard_normal((ks[b], 64)).astype(np.float32))
        items.append(Data(x=x, pos=p.cpu(), y=torch.tensor([b % 3]), edge_index=ei))
    cpu_batch = Batch.from_data_list(items)
    records, worst = _compare_model(cpu_batch, 16000, 64, dict(norm_adj=True, jk=True))
    assert records.get('gemm_128x128', 0) >= 6 and records.get('spmm_wide', 0) == 2, records
    gemm_mode.check_applied(12)


def test_full_model_exact_bench_batch_vs_oracle(gemm_mode):
    """THE benchmarked batch itself: 32 graphs (bench.py's seed 0 pool, batch 0), shipped flags, max_num_nodes = 11404 --
    forward + backward against the dense oracle in fp32 and fp64 (the oracle needs ~20 s of host time for it)."""
    ds = SyntheticCellGraphs(32, 1800, 16, base_seed=0)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(32)])
    records, worst = _compare_model(cpu_batch, 11404, 16, dict(norm_adj=True, jk=True))
    assert records.get('gemm_128x128', 0) >= 6 and records.get('spmm_wide', 0) == 2, records
    gemm_mode.check_applied(12)


def test_training_step_is_deterministic(gemm_mode):
    """Every reduction in the path has a fixed order (slot reductions, split-K combines, in-kernel accumulations): the same
    step from the same state gives BITWISE the same loss and gradients, run after run."""
    ds = SyntheticCellGraphs(8, 1800, 16, base_seed=3)
    batch = Batch.from_data_list([ds[i] for i in range(8)]).to(DEV)
    torch.manual_seed(1)
    model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                          norm_adj=True, jk=True, drop_out=0.).to(DEV)
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for _ in range(3):
        model.load_state_dict(state)
        model.zero_grad()
        logits, loss = model(batch)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    for other in runs[1:]:
        assert torch.equal(runs[0][0], other[0]) and torch.equal(runs[0][1], other[1])
        for k in runs[0][2]:
            assert torch.equal(runs[0][2][k], other[2][k]), k
