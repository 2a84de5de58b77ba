"""GPU: the step sequencer (cgc_level_fwd / cgc_level_bwd, csrc/exec.hip, native.py) against the per-operator path (ops.py).

Both enqueue the same kernels on the same operands, so logits, loss, assignment matrices, every parameter gradient and the
BatchNorm buffers must agree BIT FOR BIT wherever the two schedules are the same arithmetic (the shipped widths); elsewhere to
rounding.  The per-operator path is itself held to the oracle / the reference fixtures by test_model_gpu.py."""
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import network
from cgc_net_amd.data import Batch, SyntheticCellGraphs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _pair(args, kw, seed=5, head=True):
    torch.manual_seed(seed)
    a = network.SoftPoolingGcnEncoder(*args, **kw).to(DEV)
    b = network.SoftPoolingGcnEncoder(*args, **kw).to(DEV)
    b.load_state_dict(a.state_dict())
    a.native, b.native = True, False
    a.native_head = b.native_head = head      # the fused head sums in its own (fixed) order: same head on both sides of a bitwise check
    return a.train(), b.train()


def _used_native(model, batch):
    calls = []
    orig = network.native.level

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    network.native.level = spy
    try:
        out = model(batch)
    finally:
        network.native.level = orig
    return out, len(calls)


CASES = [
    ('plain_c60', (600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(), 6, 300),
    ('shipped_c60', (600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(norm_adj=True, jk=True), 6, 300),
    ('shipped_leaky_c60', (600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(norm_adj=True, jk=True, activation='leakyrelu'), 5, 300),
    ('nobn_elu', (600, 16, 20, 20, True, False, 20, 3, 0.1, [50]), dict(norm_adj=True, activation='elu'), 4, 200),
    ('shipped_c1140_b3', (11404, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(norm_adj=True, jk=True), 3, 1800),
    ('plain_c1140_b2', (11404, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(), 2, 1500),
    ('wide_features', (1600, 64, 20, 20, True, True, 20, 3, 0.1, [50]), dict(norm_adj=True, jk=True), 3, 500),
    ('hidden16', (800, 16, 16, 16, True, True, 16, 3, 0.1, [50]), dict(norm_adj=True, jk=True), 4, 300),
]


@pytest.mark.parametrize('name,args,flags,B,nodes', CASES, ids=[c[0] for c in CASES])
def test_sequencer_equals_per_operator_path(name, args, flags, B, nodes):
    ds = SyntheticCellGraphs(2 * B, nodes, num_features=args[1], base_seed=11)
    batches = [Batch.from_data_list([ds[i] for i in range(lo, lo + B)]).to(DEV) for lo in (0, B)]
    kw = dict(concat=True, load_data_sparse=True, drop_out=0., collect_assign=True)
    kw.update(flags)
    nat, ref = _pair(args, kw)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4) for m in (nat, ref)]
    worst = 0.0
    for step in range(3):
        b = batches[step % 2]
        (ln, lossn), ncalls = _used_native(nat, b)
        assert ncalls == 3, 'the sequencer did not take all three levels'
        (lr, lossr), rcalls = _used_native(ref, b)
        assert rcalls == 0
        for o, m in zip(opts, (nat, ref)):
            o.zero_grad()
        lossn.backward()
        lossr.backward()
        pairs = [('logits', ln, lr), ('loss', lossn, lossr)]
        pairs += [('assign%d' % i, a, c) for i, (a, c) in enumerate(zip(nat.assign_matrix, ref.assign_matrix))]
        gr = dict(ref.named_parameters())
        for k, p in nat.named_parameters():
            assert p.grad is not None and gr[k].grad is not None, k
            pairs.append(('grad ' + k, p.grad, gr[k].grad))
        br = dict(ref.named_buffers())
        pairs += [('buffer ' + k, v, br[k]) for k, v in nat.named_buffers()]
        for what, x, y in pairs:
            assert x.shape == y.shape, what
            if x.dtype.is_floating_point:
                assert torch.isfinite(x).all(), what
                err = float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30)) if x.numel() else 0.0
                worst = max(worst, err)
                assert err < 2e-6, (step, what, err)
            else:
                assert torch.equal(x, y), what
        for o in opts:
            o.step()
    print('%s: worst relative difference sequencer vs per-operator path %.3g' % (name, worst))


def test_sequencer_bitwise_on_the_shipped_configuration():
    """Same kernels, same operands, same order: bit-for-bit equal at the shipped widths (H = 20)."""
    ds = SyntheticCellGraphs(4, 600, num_features=16, base_seed=3)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    nat, ref = _pair((11404, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True))
    ln, lossn = nat(b)
    lr, lossr = ref(b)
    lossn.backward()
    lossr.backward()
    assert torch.equal(ln, lr) and torch.equal(lossn, lossr)
    gr = dict(ref.named_parameters())
    diff = [k for k, p in nat.named_parameters() if not torch.equal(p.grad, gr[k].grad)]
    assert not diff, diff


def _used_native_eval(model, batch):
    calls = []
    orig = network.native.level_eval

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    network.native.level_eval = spy
    try:
        out = model(batch)
    finally:
        network.native.level_eval = orig
    return out, len(calls)


@pytest.mark.parametrize('flags', [dict(), dict(norm_adj=True, jk=True, drop_out=0.2), dict(norm_adj=True, activation='elu')],
                         ids=['plain', 'shipped', 'elu'])
def test_inference_runs_on_the_sequencer_and_equals_the_per_operator_path(flags):
    """model.eval() under no_grad -- evaluate(), train.py:21-91 -- takes the sequencer too (cgc_level_desc.eval: BatchNorm on its running
    statistics, nothing kept): three library calls per batch instead of ~260 launches issued from Python.  Logits and assignment
    matrices equal the per-operator path's to rounding (its 1 / sqrt(running_var + eps) is torch's rsqrt, the library's an IEEE
    division), the BatchNorm buffers are left alone, and the training path is untouched by a round of inference."""
    ds = SyntheticCellGraphs(8, 300, num_features=16, base_seed=19)
    train_b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    test_b = Batch.from_data_list([ds[i] for i in range(4, 8)]).to(DEV)
    kw = dict(concat=True, load_data_sparse=True, collect_assign=True)
    kw.update(flags)
    nat, ref = _pair((600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4) for m in (nat, ref)]
    for _ in range(3):                                  # move the running statistics and the weights away from their initial values
        for m, o in zip((nat, ref), opts):
            torch.manual_seed(1)
            _, loss = m(train_b)
            o.zero_grad()
            loss.backward()
            o.step()
    nat.eval(), ref.eval()
    buffers = {k: v.clone() for k, v in nat.named_buffers()}
    with torch.no_grad():
        ln, calls = _used_native_eval(nat, test_b)
        assert calls == 3, 'inference did not take the sequencer'
        an = [a.clone() for a in nat.assign_matrix]
        lr, rcalls = _used_native_eval(ref, test_b)
        assert rcalls == 0
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
    assert ln.shape == lr.shape and rel(ln, lr) < 2e-6, rel(ln, lr)
    for a, c in zip(an, ref.assign_matrix):
        assert a.shape == c.shape and rel(a, c) < 2e-6
    for k, v in nat.named_buffers():
        assert torch.equal(v, buffers[k]), k                                   # inference leaves BatchNorm's buffers alone
    lg, gcalls = _used_native_eval(nat, test_b)                                # eval mode WITH autograd: the per-operator path
    assert gcalls == 0 and rel(lg, lr) < 2e-6
    nat.train()
    (_, loss), tcalls = _used_native(nat, train_b)
    assert tcalls == 3 and torch.isfinite(loss)


def test_sequencer_falls_back_outside_its_scope():
    """GIN blocks, eval mode with gradients enabled and inputs that require a gradient stay on the per-operator path."""
    ds = SyntheticCellGraphs(3, 200, num_features=16, base_seed=1)
    b = Batch.from_data_list([ds[i] for i in range(3)]).to(DEV)
    m = network.SoftPoolingGcnEncoder(400, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True).to(DEV)
    m.eval()
    _, calls = _used_native(m, b)
    assert calls == 0
    g = network.SoftPoolingGcnEncoder(400, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                      gcn_name='GIN').to(DEV).train()
    (_, loss), calls = _used_native(g, b)
    assert calls == 0
    loss.backward()


@pytest.mark.parametrize('flags', [dict(), dict(norm_adj=True, jk=True), dict(activation='elu', jk=True)], ids=['plain', 'shipped', 'elu'])
def test_fused_head_matches_the_module_stack(flags):
    """native.head (Linear -> act -> Dropout -> Linear -> mean cross-entropy in one kernel, its backward in another) against the
    registered modules + F.cross_entropy: logits, loss and every gradient to fp32 rounding; with dropout: the mask keeps the
    expectation and the same seed gives the same mask."""
    ds = SyntheticCellGraphs(5, 300, num_features=16, base_seed=2)
    b = Batch.from_data_list([ds[i] for i in range(5)]).to(DEV)
    kw = dict(concat=True, load_data_sparse=True, drop_out=0.)
    kw.update(flags)
    nat, ref = _pair((600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw, head=True)
    ref.native = True                                  # same levels; only the head differs
    ref.native_head = False
    assert nat.native_head
    ln, lossn = nat(b)
    lr, lossr = ref(b)
    lossn.backward()
    lossr.backward()
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
    assert rel(ln, lr) < 2e-6 and rel(lossn, lossr) < 2e-6
    gr = dict(ref.named_parameters())
    for k, p in nat.named_parameters():
        if k.endswith('att.bias'):        # mathematically zero (the attention softmax is shift-invariant): absolute
            assert float(p.grad.abs().max()) < 1e-6, k
            continue
        # the head's own parameters to rounding; upstream gradients are long cancelling sums of what the head sends down (1e-4 bar)
        assert rel(p.grad, gr[k].grad) < (5e-6 if k.startswith('pred_model') else 1e-4), (k, rel(p.grad, gr[k].grad))
    # dropout: deterministic under torch.manual_seed, ~p of the hidden units dropped, survivors scaled by 1/(1-p)
    kw['drop_out'] = 0.5
    m, _ = _pair((600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw, head=True)
    outs = []
    for seed in (1, 1, 2):
        torch.manual_seed(seed)
        m.zero_grad()
        lg, ls = m(b)
        ls.backward()
        outs.append((lg.detach().clone(), m.pred_model[0].weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], outs[2][0])
    dead_rows = float((outs[0][1].abs().sum(1) == 0).float().mean())      # hidden units dropped for EVERY graph are rare; most rows live
    assert dead_rows < 0.5


def test_large_graphs_are_reordered_transparently():
    """Graphs of >= 4000 nodes are listed grid cell by grid cell inside the model (network._spatially_ordered: locality for the wide
    aggregation).  Nothing the caller sees may depend on it: logits, loss, gradients agree with the caller's order to rounding and
    the assignment matrix comes back in the caller's node order."""
    ds = SyntheticCellGraphs(2, 4600, num_features=16, base_seed=21)
    b = Batch.from_data_list([ds[i] for i in range(2)]).to(DEV)
    assert max(b._node_counts) >= network.REORDER_MIN_NODES and b.pos is not None
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, collect_assign=True)
    on, off = _pair((9000, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw)
    off.native = True
    off.reorder_large = False
    lo, losso = on(b)
    assert on._unorder is not None
    lf, lossf = off(b)
    assert off._unorder is None
    losso.backward()
    lossf.backward()
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
    assert rel(lo, lf) < 1e-5 and rel(losso, lossf) < 1e-5
    for a, c in zip(on.assign_matrix, off.assign_matrix):
        assert a.shape == c.shape and rel(a, c) < 1e-4
    gf = dict(off.named_parameters())
    for k, p in on.named_parameters():
        if k.endswith('att.bias'):
            continue
        # (a different row order is a different summation order: a handful of the ~1e7 ReLU inputs sit within fp32 rounding of zero
        # and flip, as in test_c3_permutation_invariance_forward_backward -- same 2e-3 bar; measured 5e-4)
        assert rel(p.grad, gf[k].grad) < 2e-3, (k, rel(p.grad, gf[k].grad))


@pytest.mark.parametrize('hidden', [12, 24])
def test_other_hidden_dims_stay_on_the_fused_jk_kernels(hidden):
    """--hidden-dim other than 8 / 16 / 20 (train.py): DenseJK no longer falls back to MIOpen's LSTM -- 12 runs on the matrix-core
    kernels (and through the sequencer), 24 on the thread-per-direction kernels (per-operator path).  Against the dense oracle."""
    from oracle import dense_ref
    ds = SyntheticCellGraphs(4, 250, num_features=16, base_seed=8)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(4)])
    args = (500, 16, hidden, hidden, True, True, hidden, 3, 0.1, [50])
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.)
    torch.manual_seed(4)
    ref = dense_ref.SoftPoolingGcnEncoder(*args, **kw).train()
    model = network.SoftPoolingGcnEncoder(*args, **kw)
    model.load_state_dict(ref.state_dict())
    model.to(DEV).train()
    calls = []
    orig = torch.nn.LSTM.forward
    torch.nn.LSTM.forward = lambda self, *a, **k: calls.append(1) or orig(self, *a, **k)
    try:
        (logits, loss), ncalls = _used_native(model, cpu_batch.to(DEV))
        loss.backward()
    finally:
        torch.nn.LSTM.forward = orig
    assert not calls, 'DenseJK went through torch.nn.LSTM'
    assert ncalls == (3 if hidden == 12 else 0)
    rl, rloss = ref(cpu_batch)
    rloss.backward()
    rel = lambda x, y: float((x.double().cpu() - y.double()).abs().max() / (y.double().abs().max() + 1e-3))
    assert rel(logits, rl) < 1e-4 and rel(loss, rloss) < 1e-4
    gr = dict(ref.named_parameters())
    for k, p in model.named_parameters():
        assert rel(p.grad, gr[k].grad) < 5e-4, (k, rel(p.grad, gr[k].grad))


@pytest.mark.parametrize('one_launch', [False, True])
def test_adam_equals_torch_fused_adam(one_launch):
    """cgc_net_amd.optim.Adam = torch.optim.Adam(fused=True): with the parameter lists built once, and -- given the model -- as one
    launch of cgc_adam_step on the sequencer's flat gradient buffers.  Bitwise the same parameters and optimiser state over steps,
    an LR change and a step with gradients accumulated over two backward passes (which the one-launch path must notice and leave
    to torch's kernel) included."""
    from cgc_net_amd.optim import Adam
    ds = SyntheticCellGraphs(4, 200, num_features=16, base_seed=9)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    a, c = _pair((400, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True))
    c.native = True
    oa = Adam(a.parameters(), lr=1e-3, weight_decay=1e-4, model=a if one_launch else None)
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    fast = 0
    for step in range(7):
        if step == 3:
            for o in (oa, oc):
                o.param_groups[0]['lr'] = 5e-4
        for m, o in ((a, oa), (c, oc)):
            o.zero_grad()
            for _ in range(2 if step == 4 else 1):
                _, loss = m(b)
                loss.backward()
            if o is oa and one_launch:
                fast += int(oa._fast_ready())
            o.step()
    assert fast == (5 if one_launch else 0)           # steps 1, 2, 3, 5, 6 (0 creates the state, 4 accumulates)
    for (k, p), (_, q) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(p, q), k
    sa, sc = oa.state_dict()['state'], oc.state_dict()['state']
    for i in sa:
        for key in ('step', 'exp_avg', 'exp_avg_sq'):
            assert torch.equal(sa[i][key], sc[i][key]), (i, key)


def test_one_launch_adam_checkpoint_resume():
    """state_dict() of the one-launch optimiser carries torch's per-parameter step counters (brought up to date from the host-side
    count), and a fresh optimiser that loads it continues bit for bit like the uninterrupted run."""
    from cgc_net_amd.optim import Adam
    ds = SyntheticCellGraphs(4, 200, num_features=16, base_seed=11)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    a, c = _pair((400, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True))
    c.native = True

    def run(m, o, steps):
        for _ in range(steps):
            o.zero_grad()
            _, loss = m(b)
            loss.backward()
            o.step()
    oa = Adam(a.parameters(), lr=1e-3, weight_decay=1e-4, model=a)
    run(a, oa, 5)                                              # uninterrupted: 5 steps
    oc = Adam(c.parameters(), lr=1e-3, weight_decay=1e-4, model=c)
    run(c, oc, 3)
    ck = oc.state_dict()
    assert all(float(st['step']) == 3.0 for st in ck['state'].values())
    oc2 = Adam(c.parameters(), lr=1e-3, weight_decay=1e-4, model=c)
    oc2.load_state_dict(ck)
    run(c, oc2, 2)
    for (k, p), (_, q) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(p, q), k
    assert all(float(st['step']) == 5.0 for st in oc2.state_dict()['state'].values())


def test_one_launch_adam_notices_moved_parameters():
    """The segment table holds raw parameter addresses: when the parameters move (model.to(...)) the next step must notice, fall
    back, rebuild the table and stay bit for bit on torch's trajectory."""
    from cgc_net_amd.optim import Adam
    ds = SyntheticCellGraphs(4, 200, num_features=16, base_seed=12)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    a, c = _pair((400, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True))
    c.native = True
    oa = Adam(a.parameters(), lr=1e-3, weight_decay=1e-4, model=a)
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    fast = []
    for step in range(6):
        if step == 3:
            for m in (a, c):
                m.to('cpu')
                m.to(DEV)
        for m, o in ((a, oa), (c, oc)):
            o.zero_grad()
            _, loss = m(b)
            loss.backward()
            if o is oa:
                fast.append(bool(oa._fast_ready()))
            o.step()
    assert fast == [False, True, True, False, True, True]
    for (k, p), (_, q) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(p, q), k


def test_one_launch_adam_gradient_scale():
    """grad_mul (the 1 / replicas of a summing all-reduce folded into the optimiser): the one-launch kernel scales the gradient before
    the weight decay is added, exactly as multiplying the gradients beforehand does."""
    from cgc_net_amd.optim import Adam
    ds = SyntheticCellGraphs(4, 200, num_features=16, base_seed=13)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    a, c = _pair((400, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True))
    c.native = True
    oa = Adam(a.parameters(), lr=1e-3, weight_decay=1e-4, model=a, grad_mul=0.25)
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    used = 0
    for step in range(4):
        for m, o in ((a, oa), (c, oc)):
            o.zero_grad()
            _, loss = m(b)
            loss.backward()
            if o is oc:
                torch._foreach_mul_([p.grad for p in c.parameters()], 0.25)
            else:
                used += int(oa._fast_ready())
            o.step()
    assert used == 3
    for (k, p), (_, q) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(p, q), k


def test_composite_graph_build_equals_the_four_calls():
    from cgc_net_amd import kernels
    K = kernels.get()
    ds = SyntheticCellGraphs(3, 400, num_features=16, base_seed=4)
    b = Batch.from_data_list([ds[i] for i in range(3)]).to(DEV)
    n = b.x.shape[0]
    for p in (None, 0.4):
        one = K.graph_build(b.edge_index, n, p)
        s = K.csr_build(b.edge_index, n, add_diag=p is not None)
        nnz = int(s['rowptr'][n])
        for k in ('rowptr', 't_rowptr'):
            assert torch.equal(one[k], s[k]), k
        for k in ('col', 'rowidx', 't_col', 't_perm'):
            assert torch.equal(one[k][:nnz], s[k][:nnz]), k
        val = t_val = None
        if p is not None:
            val = torch.empty(s['cap'], device=DEV)
            K.edge_renorm(s['rowptr'], s['col'], n, p, val)
            t_val = torch.empty_like(val)
            K.csr_transpose_vals(s['t_rowptr'], s['t_perm'], val, n, t_val)
            assert torch.equal(one['val'][:nnz], val[:nnz]) and torch.equal(one['t_val'][:nnz], t_val[:nnz])
        inv = torch.empty(n, device=DEV)
        K.csr_invdeg(s['rowptr'], val, n, inv)
        assert torch.equal(one['inv_d'][:n], inv) and int(one['bad_edges']) == 0


class _FixedMask(torch.nn.Module):
    """Stands in for nn.Dropout with the keep-scale plane the fused head actually applied (0 or 1/(1-p) per element)."""

    def __init__(self, keep):
        super().__init__()
        self.keep = keep

    def forward(self, h):
        return h * self.keep.to(h.dtype)


@pytest.mark.parametrize('p', [0.2, 0.5])
def test_fused_head_dropout_is_mask_exact(p):
    """The benchmarked arithmetic (parallel_train.sh:3 ``--drop 0.2``; model/network.py:230-231 nn.Dropout between the head's two
    Linear layers), checked where it differs from drop_out = 0: the keep plane of the fused head (csrc/head.hip ``keep``) is read
    back and (1) holds exactly 0 or fp32 1/(1-p), (2) keeps a fraction 1-p of the B x 50 hidden units within a 4-sigma binomial
    band, (3) applied as a fixed mask to the registered module stack AND to the CPU oracle reproduces logits, loss and every
    gradient, (4) the backward is zero exactly where the forward was dropped."""
    from cgc_net_amd import native
    from oracle import dense_ref
    B = 32
    ds = SyntheticCellGraphs(B, 60, num_features=16, base_seed=31)
    cpu_batch = Batch.from_data_list([ds[i] for i in range(B)])
    b = cpu_batch.to(DEV)
    args = (128, 16, 20, 20, True, True, 20, 3, 0.1, [50])
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=p)
    nat, ref = _pair(args, kw, head=True)
    ref.native, ref.native_head = True, False          # same levels (bitwise); the head through the registered modules
    assert isinstance(nat.pred_model[2], torch.nn.Dropout) and nat.pred_model[2].p == p
    torch.manual_seed(7)
    ln, lossn = nat(b)
    keep = native.last_dropout_mask(nat).clone()
    lossn.backward()
    # (1) values, (2) kept fraction
    scale = (torch.ones((), dtype=torch.float32) / (torch.ones((), dtype=torch.float32) - torch.tensor(p, dtype=torch.float32))).item()
    assert keep.shape == (B, 50)
    vals = set(keep.unique().tolist())
    assert vals == {0.0, scale}, (vals, scale)
    n = keep.numel()
    frac = float((keep != 0).float().mean())
    assert abs(frac - (1 - p)) <= 4 * (p * (1 - p) / n) ** 0.5, (frac, p)
    # (3a) the module stack with THIS mask
    ref.pred_model[2] = _FixedMask(keep)
    lr, lossr = ref(b)
    lossr.backward()
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
    assert rel(ln, lr) < 2e-6 and rel(lossn, lossr) < 2e-6, (rel(ln, lr), rel(lossn, lossr))
    gr = dict(ref.named_parameters())
    for k, q in nat.named_parameters():
        if k.endswith('att.bias'):
            assert float(q.grad.abs().max()) < 1e-6, k
            continue
        assert rel(q.grad, gr[k].grad) < (5e-6 if k.startswith('pred_model') else 1e-4), (k, rel(q.grad, gr[k].grad))
    # (3b) the CPU oracle (the reference's algorithm) with THIS mask
    ora = dense_ref.SoftPoolingGcnEncoder(*args, **dict(kw, gcn_name='SAGE'))
    ora.load_state_dict({k: v.cpu() for k, v in nat.state_dict().items()})     # (weights unchanged so far: no optimiser step;
    # the BatchNorm buffers advanced by one forward do not enter a training-mode forward)
    ora.train()
    ora.pred_model[2] = _FixedMask(keep.cpu())
    ol, oloss = ora(cpu_batch)
    oloss.backward()
    assert rel(ln.cpu(), ol) < 1e-4 and rel(lossn.cpu(), oloss) < 1e-4
    go = dict(ora.named_parameters())
    for k, q in nat.named_parameters():
        if k.endswith('att.bias'):
            continue
        assert rel(q.grad.cpu(), go[k].grad) < 5e-4, (k, rel(q.grad.cpu(), go[k].grad))
    # (4) one graph: d loss / d b1[j] IS the hidden unit's upstream gradient -- exactly zero where the unit was dropped (and the whole
    # row j of d W1 with it), non-zero for kept units with a positive pre-activation
    one = Batch.from_data_list([ds[0]]).to(DEV)
    nat.zero_grad()
    torch.manual_seed(11)
    _, l1 = nat(one)
    k1 = native.last_dropout_mask(nat).clone()[0]
    ws, _, H1 = nat.__dict__['_last_head']
    z = ws[:H1].clone()
    l1.backward()
    db1, dW1 = nat.pred_model[0].bias.grad, nat.pred_model[0].weight.grad
    dropped = k1 == 0
    assert bool(dropped.any()) and bool((~dropped).any())
    assert bool((db1[dropped] == 0).all()) and bool((dW1[dropped] == 0).all())
    live = (~dropped) & (z > 0)
    assert bool(live.any()) and bool((db1[live] != 0).all())


def test_one_launch_adam_skips_a_parameter_frozen_mid_training():
    """torch.optim.Adam does not touch a parameter without a gradient.  The sequencer still writes a frozen parameter's slice of the
    flat gradient buffer, so the one-launch path must notice (p.grad is None after zero_grad()) and leave the step to torch's
    kernel: bitwise torch's trajectory, frozen parameter and its moments unchanged."""
    from cgc_net_amd.optim import Adam
    ds = SyntheticCellGraphs(4, 200, num_features=16, base_seed=14)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    a, c = _pair((400, 16, 20, 20, True, True, 20, 3, 0.1, [50]), dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True))
    c.native = True
    oa = Adam(a.parameters(), lr=1e-3, weight_decay=1e-4, model=a)
    oc = torch.optim.Adam(c.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    fast, frozen_before = [], None
    for step in range(8):
        if step == 3:
            for m in (a, c):
                m.GCN_embed_2.gcn2.weight.requires_grad_(False)
            frozen_before = a.GCN_embed_2.gcn2.weight.detach().clone()
        if step == 5:
            for m in (a, c):
                m.GCN_embed_2.gcn2.weight.requires_grad_(True)
        for m, o in ((a, oa), (c, oc)):
            o.zero_grad()
            _, loss = m(b)
            loss.backward()
            if o is oa:
                fast.append(bool(oa._fast_ready()))
            o.step()
        if step == 4:
            assert torch.equal(a.GCN_embed_2.gcn2.weight, frozen_before)
    # (after the thaw the parameters' step counts differ -- one sat out two steps --, so torch's per-parameter bias corrections
    # apply and the one-count kernel must stay out: oa._uneven)
    assert fast[:6] == [False, True, True, False, False, True] and oa._uneven, fast
    for (k, p), (_, q) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(p, q), k


def test_encoder_survives_deepcopy_and_pickle_after_native_steps():
    """EMA / best-model snapshots: copy.deepcopy(model) and torch.save(model) after training steps on the sequencer (whose
    per-encoder caches hold ctypes pointer structs) and after DataParallel-style static_flat()."""
    import copy
    import io
    from cgc_net_amd import native
    ds = SyntheticCellGraphs(3, 150, num_features=16, base_seed=15)
    b = Batch.from_data_list([ds[i] for i in range(3)]).to(DEV)
    m = network.SoftPoolingGcnEncoder(300, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True, norm_adj=True,
                                      jk=True, drop_out=0.2).to(DEV).train()
    assert native.static_flat(m) is not None
    _, loss = m(b)
    loss.backward()
    assert '_native_prepared' in m.__dict__
    snap = copy.deepcopy(m)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    outs = []
    for other in (snap, loaded):
        assert '_native_prepared' not in other.__dict__
        for (k, p), (_, q) in zip(m.state_dict().items(), other.state_dict().items()):
            assert torch.equal(p, q), k
    for other in (snap, loaded, m):                    # the copies run on the sequencer again and compute what the original computes
        torch.manual_seed(3)
        (lo, losso), calls = _used_native(other.train(), b)
        assert calls == 3
        outs.append((lo.detach().clone(), losso.detach().clone()))
    for lo, losso in outs[:2]:
        assert torch.equal(lo, outs[2][0]) and torch.equal(losso, outs[2][1])


def test_fused_head_ignores_labels_like_cross_entropy():
    """F.cross_entropy (model/network.py:289) skips labels equal to ignore_index = -100 and averages over the others; the fused
    head must do the same (it used to index its logits with the raw label)."""
    B = 6
    ds = SyntheticCellGraphs(B, 80, num_features=16, base_seed=41)
    items = [ds[i] for i in range(B)]
    b = Batch.from_data_list(items).to(DEV)
    b.y = b.y.clone()
    b.y.view(-1)[1] = -100
    b.y.view(-1)[4] = -100
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.)
    nat, ref = _pair((160, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw, head=True)
    ref.native, ref.native_head = True, False
    ln, lossn = nat(b)
    lr, lossr = ref(b)
    lossn.backward()
    lossr.backward()
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
    assert torch.isfinite(lossn) and rel(ln, lr) < 2e-6 and rel(lossn, lossr) < 2e-6
    gr = dict(ref.named_parameters())
    for k, q in nat.named_parameters():
        if k.startswith('pred_model'):
            assert rel(q.grad, gr[k].grad) < 5e-6, (k, rel(q.grad, gr[k].grad))


@pytest.mark.parametrize('frozen', ['GCN_embed_2', 'GCN_pool_1', 'GCN_pool_1.bn2'])
def test_a_frozen_block_during_training_takes_the_per_operator_path(frozen):
    """Fine-tuning with one block (or one BatchNorm) in eval() while the encoder trains: the sequencer takes ONE train / eval decision
    per level, so such a level must run on the per-operator path -- running statistics for the frozen module, batch statistics for the
    others, its buffers untouched -- and backward must work (it used to fail with CGC_EINVAL or silently use batch statistics)."""
    ds = SyntheticCellGraphs(4, 200, num_features=16, base_seed=29)
    b = Batch.from_data_list([ds[i] for i in range(4)]).to(DEV)
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.)
    nat, ref = _pair((600, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw)
    for m in (nat, ref):
        m(b)                                            # one training step's worth of running statistics
        mod = m
        for part in frozen.split('.'):
            mod = getattr(mod, part)
        mod.eval()
    level = int(frozen.split('.')[0][-1])
    before = {k: v.clone() for k, v in nat.state_dict().items() if frozen in k and 'running' in k}
    (ln, lossn), calls = _used_native(nat, b)
    assert calls == 2                                   # the two other levels stay on the sequencer, level %d does not % level
    lr, lossr = ref(b)
    lossn.backward()
    lossr.backward()
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
    assert rel(ln, lr) < 1e-5 and rel(lossn, lossr) < 1e-5
    for k, v in nat.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k]), k          # the frozen module's running statistics did not move
    gr = dict(ref.named_parameters())
    for k, q in nat.named_parameters():
        if not k.endswith('att.bias'):
            assert rel(q.grad, gr[k].grad) < 2e-3, (k, rel(q.grad, gr[k].grad))


def test_fused_head_poisons_the_step_on_a_corrupt_label():
    """torch's F.cross_entropy (model/network.py:289) raises a device-side assertion for a label outside [0, L) that is not the
    ignore value -100.  A kernel cannot raise: the fused head makes the loss and every gradient NaN instead of silently dropping the
    sample from the mean (and never uses the label as an index)."""
    B = 4
    ds = SyntheticCellGraphs(B, 80, num_features=16, base_seed=43)
    b = Batch.from_data_list([ds[i] for i in range(B)]).to(DEV)
    kw = dict(concat=True, load_data_sparse=True, norm_adj=True, jk=True, drop_out=0.)
    for bad in (3, -1, 7):
        nat, _ = _pair((160, 16, 20, 20, True, True, 20, 3, 0.1, [50]), kw, head=True)
        b.y = b.y.clone()
        b.y.view(-1)[2] = bad
        logits, loss = nat(b)
        assert torch.isfinite(logits).all() and torch.isnan(loss), (bad, loss)
        loss.backward()
        assert torch.isnan(nat.pred_model[0].weight.grad).all()
