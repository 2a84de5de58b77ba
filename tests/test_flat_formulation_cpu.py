"""CPU: the product's modules + autograd layer, driven by the torch restatement of the kernel contract,
must reproduce the reference-generated golden fixtures (forward, loss, every gradient, 3 Adam steps,
BatchNorm buffers, eval logits).  This pins the flat/CSR formulation and the hand-derived backward
schedules; the HIP kernels are then checked op by op against the same restatement on the GPU."""
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import network
from util import CASES, build_model, load_case, rel_err

TOL = 1e-4        # outputs: the north-star tolerance (fp32, relative)
# Gradients: the reference's OWN fp32 gradients sit up to 8e-5 (relative) away from an fp64 evaluation of the
# same network on the medium fixtures (measured: GCN_embed_3.gcn1.bias), and a second fp32 evaluation order is
# as far on its own -- so two correct fp32 implementations can differ by ~2e-4.  5e-4 is the gradient tolerance.
TOL_GRAD = 5e-4


@pytest.mark.parametrize('name', CASES)
def test_golden_forward_backward(name, torch_kernels):
    cfg, batch, sd, out, grad, sd3 = load_case(name)
    model = build_model(network.SoftPoolingGcnEncoder, cfg, collect_assign=True)
    model.load_state_dict(sd, strict=True)
    model.train()
    logits, loss = model(batch)
    assert rel_err(logits, out['logits']) < TOL
    assert rel_err(loss, out['loss']) < TOL
    for i, s in enumerate(model.assign_matrix):
        assert s.shape == out['assign%d' % (i + 1)].shape
        assert rel_err(s, out['assign%d' % (i + 1)]) < TOL
    loss.backward()
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, grad[k]) < TOL_GRAD, k


@pytest.mark.parametrize('name', CASES)
def test_golden_three_adam_steps(name, torch_kernels):
    cfg, batch, sd, out, grad, sd3 = load_case(name)
    model = build_model(network.SoftPoolingGcnEncoder, cfg)
    model.load_state_dict(sd, strict=True)
    model.train()
    _, loss = model(batch)          # the fixture's recorded sequence: 1 fwd/bwd, then 3 full steps
    loss.backward()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    for _ in range(3):
        _, loss = model(batch)
        opt.zero_grad()
        torch.mean(loss).backward()
        opt.step()
    for k, v in model.state_dict().items():
        # Adam's g/sqrt(v) amplifies rounding differences of near-zero gradients: 2e-3 on stepped weights
        tol = 2e-3 if v.dtype.is_floating_point else 0
        if v.dtype.is_floating_point:
            assert rel_err(v, sd3[k]) < tol, k
        else:
            assert int(v) == int(sd3[k]), k
    model.eval()
    with torch.no_grad():
        logits = model(batch)
    assert rel_err(logits, out['eval_logits3']) < 5e-3


@pytest.mark.parametrize('flags', [dict(gcn_name='GIN'), dict(gcn_name='GIN', norm_adj=True, activation='elu'),
                                   dict(activation='leakyrelu', jk=True)])
def test_variants_vs_dense_oracle(flags, torch_kernels):
    """Model variants the fixtures do not cover (GIN convolution, leaky ReLU), against the dense oracle directly."""
    from cgc_net_amd.data import Batch, SyntheticCellGraphs
    from oracle import dense_ref
    ds = SyntheticCellGraphs(4, 60, num_features=6, base_seed=13)
    batch = Batch.from_data_list([ds[i] for i in range(4)])
    args = (120, 6, 8, 8, True, True, 8, 3, 0.2, [50])
    kw = dict(concat=True, load_data_sparse=True, drop_out=0.)
    kw.update(flags)
    torch.manual_seed(5)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw)
    model = network.SoftPoolingGcnEncoder(*args, **kw)
    model.load_state_dict(ref.state_dict())
    model.train()
    ref.train()
    logits, loss = model(batch)
    rl, rloss = ref(batch)
    assert rel_err(logits, rl) < TOL and rel_err(loss, rloss) < TOL
    loss.backward()
    rloss.backward()
    gref = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        assert rel_err(p.grad, gref[k].grad) < TOL_GRAD, k


def test_wide_assignment_rows_keep_a_padded_stride(torch_kernels):
    """Cluster counts >= 256 that are not multiples of 32 (the reference's 1140): the assignment / aggregation tensors are
    strided views over rows padded to 32 floats (ops._wide); results and gradients must not notice."""
    from cgc_net_amd import ops
    from cgc_net_amd.data import Batch, SyntheticCellGraphs
    from oracle import dense_ref
    assert ops._wide(5, 300, 'cpu').stride(0) == 320 and ops._wide(5, 256, 'cpu').stride(0) == 256
    ds = SyntheticCellGraphs(3, 40, num_features=6, base_seed=3)
    batch = Batch.from_data_list([ds[i] for i in range(3)])
    args = (2700, 6, 8, 8, True, True, 8, 3, 0.1, [50])            # C1 = 270 (padded to 288), C2 = 27
    kw = dict(concat=True, load_data_sparse=True, drop_out=0., norm_adj=True, jk=True, collect_assign=True)
    torch.manual_seed(2)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw)
    model = network.SoftPoolingGcnEncoder(*args, **kw)
    model.load_state_dict(ref.state_dict())
    model.train()
    ref.train()
    logits, loss = model(batch)
    rl, rloss = ref(batch)
    assert rel_err(logits, rl) < TOL and rel_err(loss, rloss) < TOL
    assert model.assign_matrix[0].shape == ref.assign_matrix[0].shape
    assert rel_err(model.assign_matrix[0], ref.assign_matrix[0]) < TOL
    loss.backward()
    rloss.backward()
    gref = dict(ref.named_parameters())
    for k, p in model.named_parameters():      # 40-node graphs spread over 270 clusters: fp32 noise of either side ~1e-3
        assert rel_err(p.grad, gref[k].grad) < 2e-3, k


def test_dense_tuple_input_and_operator_contracts(torch_kernels):
    """The tuple input form (model/network.py:253-256) and the DenseSAGEConv / GNN_Module dense contracts with mask."""
    from cgc_net_amd.data import Batch, SyntheticCellGraphs
    from oracle import dense_ref
    ds = SyntheticCellGraphs(3, 40, num_features=6, base_seed=9)
    b = Batch.from_data_list([ds[i] for i in range(3)])
    adj = dense_ref.to_dense_adj(b.edge_index, b.batch)
    x, counts = dense_ref.to_dense_batch(b.x, b.batch)
    args = (80, 6, 8, 8, True, True, 8, 3, 0.2, [50])
    torch.manual_seed(1)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, load_data_sparse=False)
    model = network.SoftPoolingGcnEncoder(*args, load_data_sparse=False)
    model.load_state_dict(ref.state_dict())
    for train in (True, False):
        model.train(train)
        ref.train(train)
        want, got = ref((x, adj, counts, b.y)), model((x, adj, counts, b.y))
        if train:
            assert rel_err(got[0], want[0]) < TOL and rel_err(got[1], want[1]) < TOL
        else:
            assert rel_err(got, want) < TOL
    mask = dense_ref.node_mask(adj.shape[1], counts)
    rc, pc = dense_ref.DenseSAGEConv(6, 5), network.DenseSAGEConv(6, 5)
    pc.load_state_dict(rc.state_dict())
    for add_loop in (True, False):
        for m in (None, mask):
            assert rel_err(pc(x, adj, m, add_loop), rc(x, adj, m, add_loop)) < TOL
    rb, pb = dense_ref.GNNBlock(6, 8, 5, lin=True), network.GNN_Module(6, 8, 5, True, True, False, lin=True)
    pb.load_state_dict(rb.state_dict())
    xg, xr = x.clone().requires_grad_(), x.clone().requires_grad_()
    ag, ar = adj.clone().requires_grad_(), adj.clone().requires_grad_()
    got, want = pb(xg, ag, mask), rb(xr, ar, mask)
    assert rel_err(got, want) < TOL
    w = torch.randn_like(want)
    (got * w).sum().backward()
    (want * w).sum().backward()
    assert rel_err(xg.grad, xr.grad) < TOL_GRAD and rel_err(ag.grad, ar.grad) < TOL_GRAD


def _padded_dense_case(pad, seed=9, all_negative=True):
    """Dense-tuple inputs whose padding N exceeds the largest graph (the reference's dense loader pads to a FIXED
    max_num_nodes, dataflow/data.py:234,268): the BatchNorm row count is B*N and every graph's readout competes with zero rows."""
    from cgc_net_amd.data import Batch, SyntheticCellGraphs
    from oracle import dense_ref
    ds = SyntheticCellGraphs(3, 40, num_features=6, base_seed=seed)
    b = Batch.from_data_list([ds[i] for i in range(3)])
    adj = dense_ref.to_dense_adj(b.edge_index, b.batch)
    x, counts = dense_ref.to_dense_batch(b.x, b.batch)
    N = adj.shape[1] + pad
    adj_p = torch.zeros(3, N, N)
    adj_p[:, :adj.shape[1], :adj.shape[1]] = adj
    x_p = torch.zeros(3, N, x.shape[2])
    x_p[:, :x.shape[1]] = x
    return b, x_p, adj_p, counts


def test_dense_tuple_padded_beyond_largest_graph(torch_kernels):
    from oracle import dense_ref
    b, x, adj, counts = _padded_dense_case(pad=7)
    args = (80, 6, 8, 8, True, True, 8, 3, 0.2, [50])
    torch.manual_seed(1)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, load_data_sparse=False, collect_assign=True)
    model = network.SoftPoolingGcnEncoder(*args, load_data_sparse=False, collect_assign=True)
    model.load_state_dict(ref.state_dict())
    model.train(), ref.train()
    (rl, rloss), (gl, gloss) = ref((x, adj, counts, b.y)), model((x, adj, counts, b.y))
    assert rel_err(gl, rl) < TOL and rel_err(gloss, rloss) < TOL
    assert model.assign_matrix[0].shape == ref.assign_matrix[0].shape == (3, x.shape[1], 16)
    assert rel_err(model.assign_matrix[0], ref.assign_matrix[0]) < TOL
    rloss.backward(), gloss.backward()
    gref = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        assert rel_err(p.grad, gref[k].grad) < TOL_GRAD, k
    rbuf = dict(ref.named_buffers())
    for k, a in model.named_buffers():          # running statistics with count = B*N (padding included)
        if a.dtype.is_floating_point:
            assert rel_err(a, rbuf[k]) < TOL, k
    model.eval(), ref.eval()
    assert rel_err(model((x, adj, counts)), ref((x, adj, counts))) < TOL


@pytest.mark.parametrize('flags,lo,hi', [(dict(), 0.0, 1e-4), (dict(norm_adj=True, jk=True), 0.0, 1e-4),
                                         (dict(gcn_name='GIN'), 5e-4, 3e-3)], ids=['plain', 'shipped', 'GIN'])
def test_fp32_rounding_spread_of_the_reference_algorithm(flags, lo, hi):
    """What the gradient tolerances of the parity tests rest on: the REFERENCE's own algorithm (dense oracle) evaluated in
    fp32 vs in fp64 on the same input and weights.  The SAGE variants (L2-normalised, mean aggregation) stay inside 1e-4 of
    the fp64 gradients, so two correct fp32 implementations may differ by up to twice that and 5e-4 (TOL_GRAD) is a real
    bound; the GIN variant (no normalisation, sum aggregation) is ill-conditioned: its fp32 gradients are already
    ~1.3e-3 away from fp64, which is why tests/test_model_gpu.py holds GIN to 3e-3 and nothing tighter is meaningful."""
    import copy
    from cgc_net_amd.data import Batch, SyntheticCellGraphs
    from oracle import dense_ref
    ds = SyntheticCellGraphs(6, 300, num_features=16, base_seed=42)
    b = Batch.from_data_list([ds[i] for i in range(6)])
    adj = dense_ref.to_dense_adj(b.edge_index, b.batch)
    x, counts = dense_ref.to_dense_batch(b.x, b.batch)
    kw = dict(concat=True, load_data_sparse=False, drop_out=0.)
    kw.update(flags)
    torch.manual_seed(3)
    m32 = dense_ref.SoftPoolingGcnEncoder(600, 16, 20, 20, True, True, 20, 3, 0.1, [50], **kw)
    m64 = copy.deepcopy(m32).double()
    m32.train(), m64.train()
    l32, s32 = m32((x, adj.clone(), counts, b.y))
    l64, s64 = m64((x.double(), adj.double(), counts, b.y))
    assert l64.dtype == torch.float64
    s32.backward(), s64.backward()
    assert rel_err(l32, l64) < 1e-5
    g64 = dict(m64.named_parameters())
    spread = max(rel_err(p.grad, g64[k].grad) for k, p in m32.named_parameters())
    assert lo <= spread < hi, spread
