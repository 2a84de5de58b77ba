"""GPU: the full hot path (SoftPoolingGcnEncoder forward + backward through the HIP kernels) against
(1) the reference-generated golden fixtures and (2) the dense CPU oracle on seeded synthetic cell graphs."""
import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import kernels, network
from cgc_net_amd.data import Batch, SyntheticCellGraphs
from oracle import dense_ref
from util import CASES, build_model, elementwise_excess, load_case, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4
# THE GRADIENT CONTRACT IS 1e-4 STRICT AGAINST FLOAT64 (max|a-b| / max|b|, no absolute slack): the reference's float64 fixtures
# (test_golden_gradients_within_1e4_of_the_reference_in_fp64*, all eight cases, default routing and the forced big route in both GEMM
# modes) and the oracle in float64 (test_synthetic_cell_graphs_gradients_within_1e4_of_fp64, tests/test_bench_size_parity_gpu.py).
# SANITY_GRAD_FP32 is NOT that contract: it is what two fp32 evaluations of the same network (HIP path, fp32 oracle) can be held to
# against each other -- each is 2-3e-5 .. 3e-4 from the truth on these inputs (test_fp32_rounding_spread_of_the_reference_algorithm) --
# and is used only by the module-level tests below that have no float64 yardstick (dense tuple input form, single operators).
SANITY_GRAD_FP32 = 5e-4


def strict(a, b):
    """max|a-b| / max|b|: no absolute slack (util.rel_err adds 1e-3 to the denominator)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _reference_fp32_distance(name):
    """{parameter: max-norm relative distance between the reference's OWN fp32 gradient (tests/golden/<name>.npz) and its fp64
    gradient (<name>_fp64.npz)} -- both fixtures were produced by the imported reference."""
    import discrete
    fix = discrete.load_reference_fp64(name)
    _, _, _, _, grad, _ = load_case(name)
    out = {}
    for k, g in grad.items():
        m = float(fix['grad'][k].abs().max())
        out[k] = float((g.double() - fix['grad'][k]).abs().max()) / m if m > 1e-12 else 0.0
    return out


FP32_FIXTURE_BAR_CAP = 7e-4     # what round 3 held every parameter to; the data-derived bar below may be tighter, never looser
# The '*_plain' fixtures (no feature normalisation: coarsened clusters with identical content) hold ~10 / ~100 readout maxima that
# fp32 cannot decide (tests/discrete.py; profiles/r06_discrete_decisions.txt: every evaluation -- the reference's fp32 one, ours in each
# GEMM mode -- takes 92-98 of medium_plain's winners differently from float64, a different subset each).  Two fp32 evaluations of such
# a fixture differ by that choice, "up to ~1e-3 of the gradient's max-norm" (discrete.py), whatever their arithmetic: measured on the
# forced big route 3.1e-4 (exact), 3.2e-4 (six bf16 pairs), 7.0e-4 / 7.8e-4 (three fp16 pairs) -- while against the float64 fixture
# WITH the decisions routed (the contract, next test) the three modes stand at 7.0e-5 / 6.1e-5 / 5.9e-5.  The fp32-to-fp32 gradient
# check of these two fixtures is therefore held to the decision-noise magnitude, not to 2.5 x one sample of it.
DECISION_NOISE_BAR = 1e-3


def _golden_forward_backward(name):
    """Forward (logits, loss, assignment matrices: 1e-4, max-norm AND element-wise) against the reference's fp32 fixture.  Its fp32
    GRADIENTS are a secondary check only: the contract for gradients is the float64 test (1e-4 against the reference in float64).  A
    second fp32 evaluation cannot be held closer to the reference's fp32 numbers than those are to the truth, so the bar per parameter
    is max(1e-4, 2.5 x the reference's own fp32-to-fp64 distance on that parameter, the parameter's ulp64) -- data from the two fixtures --
    capped at 7e-4
    (a regenerated fixture cannot widen it unnoticed; the computed bars are printed)."""
    cfg, batch, sd, out, grad, sd3 = load_case(name, DEV)
    model = build_model(network.SoftPoolingGcnEncoder, cfg, collect_assign=True)
    model.load_state_dict(sd)
    model.to(DEV).train()
    logits, loss = model(batch)
    assert kernels.is_native()
    assert rel_err(logits, out['logits']) < TOL and elementwise_excess(logits, out['logits'], TOL) <= 1.0
    assert rel_err(loss, out['loss']) < TOL
    assert len(model.assign_matrix) == 2
    for i, s in enumerate(model.assign_matrix):
        ref_s = out['assign%d' % (i + 1)]
        assert tuple(s.shape) == tuple(ref_s.shape)
        assert rel_err(s, ref_s) < TOL and elementwise_excess(s, ref_s, TOL) <= 1.0, i         # measured excess <= 0.25
    loss.backward()
    own = _reference_fp32_distance(name)
    import discrete
    ulp64 = discrete.load_reference_fp64(name)['ulp']      # the parameter's own conditioning (tests/discrete.py::compare_with_reference_fp64)
    bars, worst = {}, (0.0, '', 0.0, 0.0)
    for k, p in model.named_parameters():
        if k.endswith('att.bias') or float(grad[k].abs().max()) < 1e-9:   # mathematically zero (attention bias under the softmax): absolute
            assert float(p.grad.abs().max()) < 1e-6, k
            continue
        bar = min(max(1e-4, 2.5 * own[k], ulp64.get(k, 0.0)), FP32_FIXTURE_BAR_CAP)
        if name.endswith('_plain') and 2.5 * own[k] > 1e-4:          # a parameter the undecidable winners reach (its own fp32-to-fp64 distance says so)
            bar = DECISION_NOISE_BAR
        bars[k] = bar
        worst = max(worst, (strict(p.grad, grad[k]) / bar, k, strict(p.grad, grad[k]), bar))
    wide = sorted(((b, k) for k, b in bars.items() if b > 1e-4), reverse=True)
    print('%s: fp32-fixture gradient bars above 1e-4 (2.5 x the fp32-to-fp64 distance of the reference itself, cap %.0e): %s; closest to its bar: '
          '%s at %.2e of %.2e' % (name, FP32_FIXTURE_BAR_CAP, [('%.1e' % b, k) for b, k in wide[:6]], worst[1], worst[2], worst[3]))
    assert worst[0] < 1.0, worst


@pytest.mark.parametrize('name', CASES)
def test_golden_forward_backward(name):
    _golden_forward_backward(name)


@pytest.mark.parametrize('name', CASES)
def test_golden_forward_backward_on_the_big_route(name, gemm_mode, forced_big_route):
    """The same fixtures with EVERY product forced onto the 128 x 128 pipelined route, in both GEMM modes: the reference-generated
    numbers then go through k_gemm_f32<2,2,2,2> (exact) and through k_gemm_split (six bf16 MFMA pairs per fp32 product,
    csrc/gemm_split.hip) -- the kernels that carry 60 % of the benchmarked step and that no fixture reaches by itself (their products
    are far below the ~450-tile route).  Same bars as the default routing."""
    _golden_forward_backward(name)
    gemm_mode.check_applied(1)


@pytest.mark.parametrize('name', CASES)
def test_golden_gradients_within_1e4_of_the_reference_in_fp64_on_the_big_route(name, gemm_mode, forced_big_route):
    """The north-star gradient bar (1e-4 strict against the REFERENCE's float64 gradients, decisions from the same fixture) with every
    product forced onto the 128 x 128 route, exact and split: the split mode pinned to reference-produced numbers
    (profiles/r06_gradients_vs_reference_fp64_split.txt has the per-case margins)."""
    import discrete
    discrete.compare_with_reference_fp64(name)
    gemm_mode.check_applied(1)


@pytest.mark.parametrize('name', CASES)
def test_golden_gradients_within_1e4_of_the_reference_in_fp64(name):
    """The north-star bar -- 1e-4, strict -- on every parameter gradient against float64 gradients PRODUCED BY THE REFERENCE
    (tests/golden/<name>_fp64.npz: /root/reference/model/network.py imported, cast to float64, run through its dense tuple input
    form; make_golden_fp64.py), with the activation signs and readout winners taken from the same fixture (tests/discrete.py::
    compare_with_reference_fp64).  The fp32 fixtures above can only be held to 7e-4 because the reference's own fp32 evaluation is
    that far from the exact gradient on some parameters."""
    import discrete
    discrete.compare_with_reference_fp64(name)


@pytest.mark.parametrize('name', ['tiny_shipped', 'tiny_gin', 'tiny_leaky', 'tiny_tuple', 'medium_plain', 'medium_shipped'])
def test_golden_three_adam_steps(name):
    cfg, batch, sd, out, grad, sd3 = load_case(name, DEV)
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    model.load_state_dict(sd)
    model.to(DEV).train()
    _, loss = model(batch)
    loss.backward()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(3):
        _, loss = model(batch)
        opt.zero_grad()
        torch.mean(loss).backward()
        opt.step()
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            assert strict(v, sd3[k]) < 5e-4, (k, strict(v, sd3[k]))        # measured <= 2.6e-4 (was held to 2e-3 with absolute slack)
        else:
            assert int(v) == int(sd3[k]), k
    model.eval()
    with torch.no_grad():
        assert rel_err(model(batch), out['eval_logits3']) < 1e-4           # measured <= 2.4e-6 (was 5e-3)


@pytest.mark.parametrize('flags', [dict(), dict(norm_adj=True, jk=True), dict(activation='leakyrelu', norm_adj=True),
                                   dict(gcn_name='GIN')])
def test_synthetic_cell_graphs_vs_oracle(flags):
    """BASELINE config 1/2 sized graphs (~300 nodes, 16 features, k-NN edges, cluster counts 60 / 6)."""
    ds = SyntheticCellGraphs(6, 300, num_features=16, base_seed=42)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(6)])
    args = (600, 16, 20, 20, True, True, 20, 3, 0.1, [50])
    kw = dict(concat=True, load_data_sparse=True, drop_out=0.)
    kw.update(flags)
    torch.manual_seed(3)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw)
    model = network.SoftPoolingGcnEncoder(*args, **kw)
    model.load_state_dict(ref.state_dict())
    model.to(DEV).train()
    ref.train()
    logits, loss = model(cpu_batch.to(DEV))
    loss.backward()
    rl, rloss = ref(cpu_batch)
    rloss.backward()
    assert rel_err(logits, rl) < TOL and rel_err(loss, rloss) < TOL
    gref = dict(ref.named_parameters())
    # Gradients: the CONTRACT is 1e-4 against float64 -- test_synthetic_cell_graphs_gradients_within_1e4_of_fp64 below for the SAGE
    # variants, the reference-generated tiny_gin fixture (test_golden_gradients_within_1e4_of_the_reference_in_fp64) for GIN.
    if flags.get('gcn_name') == 'GIN':
        # GIN has no L2 normalisation and sums (not averages) neighbours: the network is ill-conditioned in fp32 -- the reference's
        # own fp32 gradients sit 6.3e-4 away from an fp64 evaluation on exactly this input.  The yardstick is therefore the fp64
        # evaluation of the oracle itself, no absolute slack: measured 5.3e-4 for this path, held to 1e-3 (round 2: 3e-3 vs fp32).
        import copy
        ref64 = copy.deepcopy(ref).double()
        ref64.load_data_sparse = False
        ref64.zero_grad()
        adj = dense_ref.to_dense_adj(cpu_batch.edge_index, cpu_batch.batch)
        xd, counts = dense_ref.to_dense_batch(cpu_batch.x, cpu_batch.batch)
        _, loss64 = ref64((xd.double(), adj.double(), counts, cpu_batch.y))
        loss64.backward()
        g64 = dict(ref64.named_parameters())
        for k, p in model.named_parameters():
            assert strict(p.grad, g64[k].grad) < 1e-3, (k, strict(p.grad, g64[k].grad), strict(gref[k].grad, g64[k].grad))
    # (SAGE variants: no fp32-vs-fp32 gradient assert here any more -- test_synthetic_cell_graphs_gradients_within_1e4_of_fp64 holds the
    # same configurations to 1e-4 of float64)
    rbuf = dict(ref.named_buffers())
    for k, a in model.named_buffers():
        if a.dtype.is_floating_point:
            assert rel_err(a, rbuf[k]) < TOL, k                       # BatchNorm running statistics (count = B*Nmax)


@pytest.mark.parametrize('flags', [dict(), dict(norm_adj=True, jk=True), dict(activation='leakyrelu', norm_adj=True)],
                         ids=['plain', 'shipped', 'leaky'])
def test_synthetic_cell_graphs_gradients_within_1e4_of_fp64(flags):
    """The same graphs with the gradient bar at the north-star 1e-4: against the fp64 evaluation of the oracle, undecidable
    ReLU signs / max-readout winners taken as the HIP path took them (tests/discrete.py)."""
    import discrete
    ds = SyntheticCellGraphs(6, 300, num_features=16, base_seed=42)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(6)])
    discrete.compare_model(cpu_batch, 600, 16, flags, seed=3)


def test_dense_tuple_input_form_and_eval():
    """model/network.py:253-256: (x[B,N,F], adj[B,N,N], num_nodes[, label]) input; eval mode returns logits only."""
    ds = SyntheticCellGraphs(3, 80, num_features=16, base_seed=9)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(3)])
    adj = dense_ref.to_dense_adj(cpu_batch.edge_index, cpu_batch.batch)
    x, counts = dense_ref.to_dense_batch(cpu_batch.x, cpu_batch.batch)
    args = (160, 16, 20, 20, True, True, 20, 3, 0.1, [50])
    torch.manual_seed(1)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, load_data_sparse=False)
    model = network.SoftPoolingGcnEncoder(*args, load_data_sparse=False)
    model.load_state_dict(ref.state_dict())
    model.to(DEV)
    for train in (True, False):
        model.train(train)
        ref.train(train)
        want = ref((x, adj, counts, cpu_batch.y))
        got = model((x.to(DEV), adj.to(DEV), counts, cpu_batch.y.to(DEV)))
        if train:
            assert rel_err(got[0], want[0]) < TOL and rel_err(got[1], want[1]) < TOL
        else:
            assert rel_err(got, want) < TOL


def test_dense_tuple_padded_beyond_largest_graph():
    """The dense loader pads to a fixed max_num_nodes (dataflow/data.py:234,268): N > max(counts) changes the BatchNorm
    row count (B*N), makes every readout compete with zero rows and fixes the assignment matrix's shape."""
    from test_flat_formulation_cpu import _padded_dense_case
    b, x, adj, counts = _padded_dense_case(pad=7)
    args = (80, 6, 8, 8, True, True, 8, 3, 0.2, [50])
    torch.manual_seed(1)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, load_data_sparse=False, collect_assign=True)
    model = network.SoftPoolingGcnEncoder(*args, load_data_sparse=False, collect_assign=True)
    model.load_state_dict(ref.state_dict())
    model.to(DEV).train()
    ref.train()
    rl, rloss = ref((x, adj, counts, b.y))
    gl, gloss = model((x.to(DEV), adj.to(DEV), counts, b.y.to(DEV)))
    assert rel_err(gl, rl) < TOL and rel_err(gloss, rloss) < TOL
    assert tuple(model.assign_matrix[0].shape) == tuple(ref.assign_matrix[0].shape)
    assert rel_err(model.assign_matrix[0], ref.assign_matrix[0]) < TOL
    rloss.backward(), gloss.backward()
    gref = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        assert rel_err(p.grad, gref[k].grad) < SANITY_GRAD_FP32, k
    rbuf = dict(ref.named_buffers())
    for k, a in model.named_buffers():
        if a.dtype.is_floating_point:
            assert rel_err(a, rbuf[k]) < TOL, k
    model.eval(), ref.eval()
    with torch.no_grad():
        assert rel_err(model((x.to(DEV), adj.to(DEV), counts)), ref((x, adj, counts))) < TOL


def test_operator_modules_vs_oracle():
    """DenseSAGEConv / GNN_Module dense-tensor contracts incl. mask and add_loop."""
    torch.manual_seed(0)
    B_, N, Fi, Fo = 3, 37, 10, 12
    x = torch.randn(B_, N, Fi)
    adj = (torch.rand(B_, N, N) < 0.2).float()
    counts = torch.tensor([37, 20, 5])
    mask = dense_ref.node_mask(N, counts)
    adj = adj * mask * mask.transpose(1, 2)
    x = x * mask
    rc, pc = dense_ref.DenseSAGEConv(Fi, Fo), network.DenseSAGEConv(Fi, Fo)
    pc.load_state_dict(rc.state_dict())
    pc.to(DEV)
    for add_loop in (True, False):
        for m in (None, mask):
            want = rc(x, adj, m, add_loop)
            got = pc(x.to(DEV), adj.to(DEV), None if m is None else m.to(DEV), add_loop)
            assert rel_err(got, want) < TOL
    rb = dense_ref.GNNBlock(Fi, 8, Fo, lin=True)
    pb = network.GNN_Module(Fi, 8, Fo, True, True, False, lin=True)
    pb.load_state_dict(rb.state_dict())
    pb.to(DEV).train()
    rb.train()
    xg, ag = x.to(DEV).requires_grad_(), adj.to(DEV).requires_grad_()
    xr, ar = x.clone().requires_grad_(), adj.clone().requires_grad_()
    got, want = pb(xg, ag, mask.to(DEV)), rb(xr, ar, mask)
    assert rel_err(got, want) < TOL
    w = torch.randn_like(want)
    (got * w.to(DEV)).sum().backward()
    (want * w).sum().backward()
    assert rel_err(xg.grad, xr.grad) < SANITY_GRAD_FP32 and rel_err(ag.grad, ar.grad) < SANITY_GRAD_FP32
    rgrad = {k: q.grad for k, q in rb.named_parameters()}
    for k, p in pb.named_parameters():
        assert rel_err(p.grad, rgrad[k]) < SANITY_GRAD_FP32, k


def test_training_learns_a_separable_task():
    """End-to-end sanity beyond single-step parity: with class-dependent feature means the shipped configuration (jk,
    norm_adj, dropout) fits the labels within a few dozen Adam steps on the device front-end (collate + k-NN on the GPU)."""
    from cgc_net_amd.data import Batch, Data, SyntheticCellGraphs
    from cgc_net_amd import network
    ds = SyntheticCellGraphs(48, 150, 16, base_seed=77)
    items = []
    for i in range(48):
        d = ds[i]
        x = d.x.clone()
        x[:, :4] += 1.5 * (int(d.y) - 1)                               # the signal: class shifts four feature means
        items.append(Data(x=x, pos=d.pos, y=d.y))
    torch.manual_seed(0)
    model = network.SoftPoolingGcnEncoder(300, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                          norm_adj=True, jk=True, drop_out=0.2).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, weight_decay=1e-4)
    batches = [Batch.from_data_list(items[i:i + 16], device=DEV, knn=(100.0, 8)) for i in range(0, 48, 16)]
    model.train()
    first = last = None
    for epoch in range(25):
        tot = 0.0
        for b in batches:
            _, loss = model(b)
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot += float(loss.detach())
        first = tot / 3 if first is None else first
        last = tot / 3
    model.eval()
    with torch.no_grad():
        acc = sum(int((model(b).argmax(1) == b.y.view(-1)).sum()) for b in batches) / 48.0
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    assert acc >= 0.8, acc
