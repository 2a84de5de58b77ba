"""GPU: mode CGC_GEMM_SPLIT_F16 of cgc_gemm_f32_ws / cgc_gemm_f32_cat_ws (csrc/gemm_half.hip: an fp32 product as three fp16 MFMA pairs
of operands scaled per batch item) in every form the step's dominant products take (model/network.py:121-122, 206-207), against
float64, NEXT TO the exact fp32 kernel on the same inputs.  Yardstick as in test_split_gemm_gpu.py: error of an output element relative
to sum_k |a_ik| |b_kj|.

Input families: 'normal' N(0, 1); 'tiny' (A at 2^-100: the scale must bring it back); 'items' (every batch item / the whole operand at
its own scale between 2^-40 and 2^+40); 'tiles' (flat products: every 256-row panel of op(A) and every 128-column panel of op(B) at
its own scale, 2^-40 .. 2^+40: what per-tile scales are for -- the rows of different graphs in one flat gradient tensor); 'rows12'
(output rows / columns at scales spread over 12 binades INSIDE a panel: still within the 17 binades in which both planes are normal).  Bars: rms <= 1.25 x the exact kernel's, max <= 2 x
(1.25 x from 10^6 outputs).  The analytic statement for inputs beyond that range is test_half_gemm_error_bound_with_range."""
import os

import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from test_split_gemm_gpu import DEV, EXACT, HALF, U, both_modes, hip

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def big_route(forced_big_route):
    yield


def gen(shape, seed, kind, mn=-2, item_axis=None, blk=256):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    if kind == 'tiles' and len(shape) >= 2:     # every 256-row panel of op(A) / 128-column panel of op(B) at its own scale, 2^-40 .. 2^+40
        ax = mn % len(shape)
        sh = [1] * len(shape)
        sh[ax] = shape[ax]
        nblk = -(-shape[ax] // blk)
        sc = torch.exp2(torch.randint(-40, 41, (nblk,), generator=g).float()).repeat_interleave(blk)[:shape[ax]]
        x = x * sc.view(sh)
    elif kind == 'tiny':
        x = x * 2.0 ** -100
    elif kind == 'items':
        if item_axis is None:
            x = x * 2.0 ** float(torch.randint(-40, 41, (1,), generator=g))
        else:
            sh = [1] * len(shape)
            sh[item_axis] = shape[item_axis]
            x = x * torch.exp2(torch.randint(-40, 41, sh, generator=g).float())
    elif kind == 'rows12' and len(shape) >= 2:
        ax = mn % len(shape)
        sh = [1] * len(shape)
        sh[ax] = shape[ax]
        x = x * torch.exp2(torch.randint(-6, 7, sh, generator=g).float())
    return x.to(DEV)


def check(res, what, outputs):
    (em, er), (hm, hr) = res[EXACT], res[HALF]
    line = '%s: exact max %.2e rms %.2e | half max %.2e rms %.2e  (ratios %.2f / %.2f)' % (what, em, er, hm, hr, hm / max(em, 1e-30), hr / max(er, 1e-30))
    print(line)
    path = os.environ.get('CGC_HALF_ERROR_TABLE')
    if path:
        with open(path, 'a') as fh:
            fh.write(line + '\n')
    assert hr <= 1.25 * er + 2e-9, (what, res)
    assert hm <= (1.25 if outputs >= 1000000 else 2.0) * em + 2e-9, (what, res)
    assert hm < 1e-6


KINDS = ['normal', 'tiny', 'items', 'rows12']
MODES = (EXACT, HALF)


@pytest.mark.parametrize('kind', KINDS + ['tiles'])
@pytest.mark.parametrize('M,N,K,tA,tB', [(3000, 1140, 1140, False, False), (2500, 1140, 1140, False, True), (1140, 1140, 5000, True, False),
                                         (257, 130, 170, False, False), (700, 300, 2052, False, True), (300, 260, 176, True, False)])
def test_half_gemm_flat(M, N, K, tA, tB, kind):
    k = hip()
    up4 = lambda v: (v + 3) // 4 * 4
    lda, ldb = up4(M if tA else K) + 4, up4(K if tB else N) + 4
    A = gen((K, lda) if tA else (M, lda), 1, kind, -1 if tA else -2)
    B = gen((N, ldb) if tB else (K, ldb), 2, 'normal' if kind == 'tiny' else kind, -2 if tB else -1, blk=128)
    # the padding columns behind the operands' extents belong to somebody else: they must not reach the scale
    (A[:, M:] if tA else A[:, K:]).fill_(float('nan'))
    (B[:, K:] if tB else B[:, N:]).fill_(3.0e38)
    bias, C0 = gen((N,), 3, 'normal'), gen((M, N), 4, 'normal')
    a = (A[:, :M].t() if tA else A[:, :K]).double()
    b = (B[:, :K].t() if tB else B[:, :N]).double()
    prod = a.abs() @ b.abs()
    C0, bias = (C0.double() * prod).float(), (bias.double() * prod.amin(0)).float()      # beta C at the scale of each output, the bias at its column's smallest
    want = 0.5 * (a @ b) + 2.0 * C0.double() + bias.double()
    mag = 0.5 * prod + 2.0 * C0.double().abs() + bias.double().abs()

    def run():
        out = C0.clone()
        k.gemm(A, B, out, M, N, K, tA, tB, lda, ldb, N, 0.5, 2.0, bias)
        return out
    check(both_modes(run, want, mag, MODES), 'flat %s %s' % ((M, N, K, tA, tB), kind), M * N)


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('tB,xk,beta', [(False, 0, 0.0), (True, 20, 1.0), (False, 40, 0.0)])
def test_half_gemm_ragged_m_with_extra_segment(tB, xk, beta, kind):
    """Y_b = S_b op(G_b) (+ X_b op(H_b)) (+ Y_b): a B operand per graph, the rows of graph b; 'items': every graph at its own scale."""
    k = hip()
    counts = [700, 0, 513, 128, 77, 900, 250, 640]
    n, nmax, batch, N, K = sum(counts), max(counts), len(counts), 1140, 1140
    gptr = torch.tensor(np.cumsum([0] + counts), dtype=torch.int32, device=DEV)
    gp = gptr.cpu().tolist()
    S = gen((n, K), 1, 'normal' if kind == 'items' else kind)
    X = gen((n, max(xk, 4)), 5, 'normal' if kind == 'items' else kind)
    if kind == 'items':
        g = torch.Generator(device='cpu').manual_seed(77)
        sc = torch.exp2(torch.randint(-40, 41, (batch,), generator=g).float()).to(DEV)
        rows = torch.repeat_interleave(sc, torch.tensor(counts, device=DEV))[:, None]
        S, X = S * rows, X * rows
    G = gen((batch, N, K) if tB else (batch, K, N), 2, 'normal' if kind == 'tiny' else kind, -2 if tB else -1, item_axis=0)
    H = gen((batch, N, max(xk, 4)) if tB else (batch, max(xk, 4), N), 6, 'normal', -2 if tB else -1)
    if kind == 'items':                       # (the extra segment shares its item's scale with the main one: same order of magnitude)
        H = H * (G.abs().amax(dim=(1, 2), keepdim=True))
    C0 = gen((n, N), 3, 'normal')
    op = lambda t: t.double().t() if tB else t.double()
    parts, mags = [], []
    for b in range(batch):
        s, x = S[gp[b]:gp[b + 1]].double(), X[gp[b]:gp[b + 1], :xk].double()
        h = (H[b][:, :xk] if tB else H[b][:xk]).double()
        parts.append(s @ op(G[b]) + (x @ op(h) if xk else 0))
        mags.append(s.abs() @ op(G[b]).abs() + (x.abs() @ op(h).abs() if xk else 0))
    want, mag = torch.cat(parts), torch.cat(mags)
    C0 = C0 * mag.float()                       # beta C at the scale of each output
    want, mag = want + beta * C0.double(), mag + beta * C0.double().abs()

    def run():
        out = C0.clone()
        extra = [(X, H, X.shape[1], H.shape[2], xk, 0, H.shape[1] * H.shape[2])] if xk else ()
        k.gemm(S, G, out, 0, N, K, False, tB, K, G.shape[2], N, 1.0, beta, None, batch, 0, G.shape[1] * G.shape[2], 0, gptr, 1, nmax, n,
               extra=extra)
        return out
    check(both_modes(run, want, mag, MODES), 'ragged M tB=%s xk=%d %s' % (tB, xk, kind), n * N)


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('counts', [[300, 0, 513, 128, 77, 900, 250, 640], [1800, 1900, 1750], [0, 300, 513, 128, 77, 900, 250, 640],
                                    [40, 300, 513, 128, 77, 900, 250, 640]])
def test_half_gemm_ragged_k(counts, kind):
    """out[b] = S_b^T P_b over the rows of graph b; an empty graph and a graph shorter than the tail split's pieces first."""
    k = hip()
    n, nmax, batch, C = sum(counts), max(counts), len(counts), 1140
    gptr = torch.tensor(np.cumsum([0] + counts), dtype=torch.int32, device=DEV)
    gp = gptr.cpu().tolist()
    S, P = gen((n, C), 1, 'normal' if kind == 'items' else kind, -1), gen((n, C), 2, 'normal' if kind in ('tiny', 'items') else kind, -1)
    if kind == 'items':
        g = torch.Generator(device='cpu').manual_seed(78)
        rows = lambda: torch.repeat_interleave(torch.exp2(torch.randint(-40, 41, (batch,), generator=g).float()).to(DEV), torch.tensor(counts, device=DEV))[:, None]
        S, P = S * rows(), P * rows()
    want = torch.stack([S[gp[b]:gp[b + 1]].double().t() @ P[gp[b]:gp[b + 1]].double() for b in range(batch)])
    mag = torch.stack([S[gp[b]:gp[b + 1]].double().abs().t() @ P[gp[b]:gp[b + 1]].double().abs() for b in range(batch)])
    mag = mag.clamp_min(1e-300)
    mag = torch.where(mag <= 1e-300, torch.ones_like(mag), mag)            # (an empty graph: 0 / 1)

    def run():
        out = torch.full((batch, C, C), float('nan'), device=DEV)
        k.gemm(S, P, out, C, C, 0, True, False, C, C, C, 1.0, 0.0, None, batch, 0, 0, C * C, gptr, 2, nmax, n)
        return out
    check(both_modes(run, want, mag, MODES), 'ragged K %s %s' % (counts[:3], kind), batch * C * C)


@pytest.mark.parametrize('kind', KINDS)
def test_half_gemm_uniform_k_chunks(kind):
    """ragged = 3 (the weight gradient X^T dY cut into row chunks, partial products summed afterwards): every chunk its own scale."""
    k = hip()
    M, N, Kd, chunk = 1140, 1140, 9000, 2080
    parts = -(-Kd // chunk)
    A, B = gen((Kd, M), 1, kind, -1), gen((Kd, N), 2, 'normal' if kind == 'tiny' else kind, -1)
    want = A.double().t() @ B.double()
    mag = A.double().abs().t() @ B.double().abs()

    def run():
        ws = torch.full((parts, M, N), float('nan'), device=DEV)
        k.gemm(A, B, ws, M, N, Kd, True, False, M, N, N, 1.0, 0.0, None, parts, 0, 0, M * N, None, 3, chunk, Kd)
        return ws.double().sum(0)
    check(both_modes(run, want, mag, MODES), 'uniform chunks %s' % kind, M * N)


@pytest.mark.parametrize('kind', ['normal', 'items'])
def test_half_gemm_flat_with_two_extra_segments(kind):
    """Linear over cat[x1, x2, x3] (model/network.py:118-122): a long main segment and two short ones; one scale for the three."""
    k = hip()
    M, N = 2100, 1140
    for K0 in (1140, 20):
        A, B = gen((M, K0), 1, kind), gen((K0, N), 2, kind, -1)
        sa, sb = float(A.abs().max()), float(B.abs().max())
        X1, H1 = gen((M, 20), 3, 'normal') * sa, gen((20, N), 4, 'normal', -1) * sb
        X2, H2 = gen((M, 24), 5, 'normal') * sa, gen((24, N), 6, 'normal', -1) * sb
        bias = gen((N,), 7, 'normal') * sa * sb
        want = A.double() @ B.double() + X1.double() @ H1.double() + X2.double() @ H2.double() + bias.double()
        mag = A.double().abs() @ B.double().abs() + X1.double().abs() @ H1.double().abs() + X2.double().abs() @ H2.double().abs() + bias.double().abs()

        def run():
            out = torch.full((M, N), float('nan'), device=DEV)
            k.gemm(A, B, out, M, N, K0, False, False, K0, N, N, 1.0, 0.0, bias, extra=[(X1, H1, 20, N, 20, 0, 0), (X2, H2, 24, N, 24, 0, 0)])
            return out
        check(both_modes(run, want, mag, MODES), 'cat K0=%d %s' % (K0, kind), M * N)


@pytest.mark.parametrize('tA,tB', [(False, False), (False, True), (True, False)])
def test_half_gemm_strided_batch_and_tail_split_is_deterministic(tA, tB):
    k = hip()
    batch, M, N, K = 3, 700, 600, 1333
    A = gen((batch, K, M) if tA else (batch, M, K), 1, 'items', item_axis=0)
    B = gen((batch, N, K) if tB else (batch, K, N), 2, 'items', item_axis=0)
    a = A.double().transpose(1, 2) if tA else A.double()
    b = B.double().transpose(1, 2) if tB else B.double()
    want, mag = torch.bmm(a, b), torch.bmm(a.abs(), b.abs())
    outs = []
    k.gemm_mode = HALF
    try:
        for split in (True, True, False):
            k.tail_split = split
            out = torch.full((batch, M, N), float('nan'), device=DEV)
            k.gemm(A, B, out, M, N, K, tA, tB, A.shape[2], B.shape[2], N, 1.0, 0.0, None, batch, A.shape[1] * A.shape[2], B.shape[1] * B.shape[2], M * N)
            outs.append(out)
    finally:
        k.tail_split, k.gemm_mode = True, EXACT
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    for o in outs:
        assert float(((o.double() - want).abs() / mag).max()) < 5e-7
    assert float(((outs[0].double() - outs[2].double()).abs() / mag).max()) < 5e-7


def test_half_gemm_zero_operand_and_zero_items():
    """An operand that is all zero takes scale 1 and gives exact zeros (+ beta C); so does a batch item that is."""
    k = hip()
    batch, M, N, K = 3, 300, 260, 400
    A, B = gen((batch, M, K), 1, 'normal'), gen((batch, K, N), 2, 'normal')
    A[1].zero_()
    out = torch.full((batch, M, N), float('nan'), device=DEV)
    k.gemm_mode = HALF
    try:
        k.gemm(A, B, out, M, N, K, False, False, K, N, N, 1.0, 0.0, None, batch, M * K, K * N, M * N)
    finally:
        k.gemm_mode = EXACT
    torch.cuda.synchronize()
    assert bool((out[1] == 0).all())
    want, mag = torch.bmm(A.double(), B.double()), torch.bmm(A.double().abs(), B.double().abs())
    assert float(((out.double() - want).abs()[[0, 2]] / mag[[0, 2]]).max()) < 5e-7


@pytest.mark.parametrize('spread', [8, 20, 30, 40])
def test_half_gemm_error_bound_with_range(spread):
    """Beyond the range in which both planes are normal the error is bounded, not fp32-grade: with s the item's scale (max |x| s in
    [2^14, 2^15)) an element's representation error is at most max(2^-22 |x|, 2^-24 / s) -- half a unit of the last place of l while l
    is normal, half the subnormal spacing 2^-24 of the scaled value below -- so
        |out - exact| <= sum_k (da_k |b_k| + |a_k| db_k + da_k db_k) + accumulation,
    evaluated here in float64 on operands whose scales run over 2^+-spread ALONG K in both operands (test_split_gemm_gpu's 'skewk':
    the family where a static scale must lose).  Asserted: every output within that bound (+ the accumulation allowance of the three
    pairs: 2 x 3 x ceil(K / 16) + 4 roundings of U mag); printed: how far above the exact kernel the mode actually lands."""
    k = hip()
    M, N, K = 700, 300, 1140
    g = torch.Generator(device='cpu').manual_seed(5)
    A = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-spread, spread + 1, (1, K), generator=g).float())).to(DEV)
    B = (torch.randn(K, N, generator=g) * torch.exp2(torch.randint(-spread, spread + 1, (K, 1), generator=g).float())).to(DEV)

    def rep_err(x):
        e = torch.floor(torch.log2(x.abs().max().double()))
        s = torch.exp2(14 - e)
        return torch.maximum(2.0 ** -22 * x.abs().double(), 2.0 ** -24 / s * torch.ones_like(x, dtype=torch.float64))
    a, b = A.double(), B.double()
    da, db = rep_err(A), rep_err(B)
    bound = da @ b.abs() + a.abs() @ db + da @ db
    want, mag = a @ b, a.abs() @ b.abs()
    res = {}
    for mode in (EXACT, HALF):
        out = torch.empty(M, N, device=DEV)
        k.gemm_mode = mode
        try:
            k.gemm(A, B, out, M, N, K, False, False, K, N, N)
        finally:
            k.gemm_mode = EXACT
        res[mode] = (out.double() - want).abs()
    acc = (2 * 3 * (-(-K // 16)) + 4) * U * mag
    worst = float((res[HALF] / (bound + acc)).max())
    line = ('scales of 2^+-%d along K: half max err / mag %.2e (exact kernel %.2e); err / (representation bound + accumulation allowance): max %.3f; '
            'bound / mag: median %.2e max %.2e' % (spread, float((res[HALF] / mag).max()), float((res[EXACT] / mag).max()), worst,
                                                   float((bound / mag).median()), float((bound / mag).max())))
    print(line)
    if os.environ.get('CGC_HALF_ERROR_TABLE'):
        with open(os.environ['CGC_HALF_ERROR_TABLE'], 'a') as fh:
            fh.write(line + '\n')
    assert worst <= 1.0


def test_half_mode_leaves_other_routes_exact():
    k = hip()
    k.lib.cgc_gemm_tuning(0)
    cases = [(5000, 40, 1140, False, False, 1140, 40), (3000, 1140, 100, False, True, 100, 100), (900, 700, 501, False, False, 501, 700)]
    for M, N, K, tA, tB, lda, ldb in cases:
        A, B = gen((M, lda), 1, 'normal'), gen((N if tB else K, ldb), 2, 'normal')
        outs = []
        for mode in (EXACT, HALF):
            before = int(k.lib.cgc_gemm_half_count())
            k.gemm_mode = mode
            try:
                out = torch.empty(M, N, device=DEV)
                k.gemm(A, B, out, M, N, K, tA, tB, lda, ldb, N)
            finally:
                k.gemm_mode = EXACT
            assert int(k.lib.cgc_gemm_half_count()) == before
            outs.append(out)
        assert torch.equal(outs[0], outs[1])


def test_half_mode_hands_small_products_to_the_bf16_kernel():
    """Below cgc_gemm_half_min_work tile x k-tile steps the mode's maximum pass costs more than its kernel saves: such a product runs
    in mode CGC_GEMM_SPLIT_BF16 (same route, no scaling); at or above it, on the fp16 kernel."""
    import ctypes
    k = hip()
    M, N, K = 2600, 1140, 1140              # 11 x 9 tiles x 72 k-tiles = 7128 steps
    A, B = gen((M, K), 1, 'normal'), gen((K, N), 2, 'normal', -1)
    want, mag = A.double() @ B.double(), A.double().abs() @ B.double().abs()
    old = k.lib.cgc_gemm_half_min_work(ctypes.c_int64(-1))
    try:
        for thr, to_half in ((7129, False), (7128, True)):
            assert k.lib.cgc_gemm_half_min_work(ctypes.c_int64(thr)) >= 0
            h0, s0 = int(k.lib.cgc_gemm_half_count()), int(k.lib.cgc_gemm_split_count())
            out = torch.empty(M, N, device=DEV)
            k.gemm_mode = HALF
            try:
                k.gemm(A, B, out, M, N, K, False, False, K, N, N)
            finally:
                k.gemm_mode = EXACT
            torch.cuda.synchronize()
            assert (int(k.lib.cgc_gemm_half_count()) - h0, int(k.lib.cgc_gemm_split_count()) - s0) == ((1, 0) if to_half else (0, 1))
            assert float(((out.double() - want).abs() / mag).max()) < 5e-7
    finally:
        k.lib.cgc_gemm_half_min_work(ctypes.c_int64(old))


@pytest.mark.parametrize('M,N,K,tA,tB', [(300, 140, 5, False, False), (129, 257, 1, False, True), (513, 131, 17, True, False)])
def test_half_gemm_reductions_shorter_than_the_pipeline(M, N, K, tA, tB):
    """K = 1, 5, 17: fewer k-tiles than the pipeline is deep (one or two, the last one partial), odd extents in both output directions.
    With so few terms nothing averages the representation error and the exact kernel's own error is a rounding or two, so the yardstick
    here is the mode's worst case, not the exact kernel: |s x - h - l| <= 2^-23 |s x| per operand element and the dropped pair
    l l <= 2^-22 |a||b| give 8 U of sum |a||b|, + the accumulation allowance of test_half_gemm_error_bound_with_range.  (By itself the
    dispatcher sends no product with K <= 160 to this kernel; the forced route of this file does.)"""
    k = hip()
    up4 = lambda v: (v + 3) // 4 * 4
    lda, ldb = up4(M if tA else K), up4(K if tB else N)
    A, B = gen((K, lda) if tA else (M, lda), 1, 'normal'), gen((N, ldb) if tB else (K, ldb), 2, 'normal')
    a = (A[:, :M].t() if tA else A[:, :K]).double()
    b = (B[:, :K].t() if tB else B[:, :N]).double()
    want, mag = a @ b, a.abs() @ b.abs()

    def run():
        out = torch.full((M, N), float('nan'), device=DEV)
        k.gemm(A, B, out, M, N, K, tA, tB, lda, ldb, N)
        return out
    res = both_modes(run, want, mag, MODES)
    bound = (8 + 2 * 3 * (-(-K // 16)) + 4) * U
    print('short K %s: exact max %.2e rms %.2e | half max %.2e rms %.2e | worst case %.2e' % ((M, N, K, tA, tB), res[EXACT][0], res[EXACT][1],
                                                                                               res[HALF][0], res[HALF][1], bound))
    assert res[HALF][0] <= bound and res[HALF][1] <= 0.25 * bound, (res, bound)


def test_half_gemm_non_finite_input_stays_in_its_panels():
    """The mode's domain is finite inputs; what an infinite element does is stated in the header and held here: the output tiles that
    multiply its panel (256 rows of op(A)) are not finite, every other row of the product is as accurate as without it."""
    k = hip()
    M, N, K = 1100, 300, 400
    A, B = gen((M, K), 1, 'normal'), gen((K, N), 2, 'normal', -1)
    A[300, 7] = float('inf')                   # row 300: panel 1 (rows 256 .. 511)
    out = torch.empty(M, N, device=DEV)
    k.gemm_mode = HALF
    try:
        k.gemm(A, B, out, M, N, K, False, False, K, N, N)
    finally:
        k.gemm_mode = EXACT
    torch.cuda.synchronize()
    rows = torch.ones(M, dtype=torch.bool, device=DEV)
    rows[256:512] = False
    assert not bool(torch.isfinite(out[300]).all())
    assert bool(torch.isfinite(out[rows]).all())
    want, mag = A[rows].double() @ B.double(), A[rows].double().abs() @ B.double().abs()
    assert float(((out[rows].double() - want).abs() / mag).max()) < 5e-7


def test_half_mode_without_a_workspace_runs_the_exact_kernel():
    """cgc_gemm_f32_ws with ws = NULL in mode CGC_GEMM_SPLIT_F16: no room for the scale slots -- the product runs on another kernel
    (the bf16 one needs no workspace) instead of failing; the plain entry point cgc_gemm_f32 is always exact."""
    import ctypes
    k = hip()
    M, N, K = 700, 300, 400
    A, B = gen((M, K), 1, 'normal'), gen((K, N), 2, 'normal', -1)
    want, mag = A.double() @ B.double(), A.double().abs() @ B.double().abs()
    out = torch.empty(M, N, device=DEV)
    h0 = int(k.lib.cgc_gemm_half_count())
    rc = k.lib.cgc_gemm_f32_ws(0, 0, M, N, K, ctypes.c_float(1.0), A.data_ptr(), K, B.data_ptr(), N, ctypes.c_float(0.0), out.data_ptr(), N, None, 1,
                               ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), None, 0, 0, None, ctypes.c_int64(0), int(HALF),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and int(k.lib.cgc_gemm_half_count()) == h0
    assert float(((out.double() - want).abs() / mag).max()) < 5e-7
    assert k.lib.cgc_gemm_f32_ws(0, 0, M, N, K, ctypes.c_float(1.0), A.data_ptr(), K, B.data_ptr(), N, ctypes.c_float(0.0), out.data_ptr(), N, None, 1,
                                 ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0), None, 0, 0, None, ctypes.c_int64(0), 3,
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0          # an unknown mode is refused
