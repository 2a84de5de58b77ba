"""GPU: mode CGC_GEMM_SPLIT_BF16 of cgc_gemm_f32_ws / cgc_gemm_f32_cat_ws (csrc/gemm_split.hip: an fp32 product as six bf16 MFMA
pairs) in every form the step's dominant products take -- the assignment Linear and _diff_pool's contractions with their backward
(model/network.py:121-122, 206-207) -- against float64, NEXT TO the exact fp32 kernel on the same inputs.

Yardstick: error of an output element relative to sum_k |a_ik| |b_kj| (the quantity fp32 rounding scales with).  Bars: maximum and
rms of the split mode <= 1.25 x those of the exact kernel (+ 2e-9), on N(0, 1) inputs, on inputs whose output rows / columns carry
scales from 2^-30 to 2^+30, and on inputs near the bottom of the exponent range (2^-100); a fourth input family puts the scales along
K (see check())."""
import os

import numpy as np
import pytest
import torch

import cgc_net_amd  # noqa: F401
from cgc_net_amd import kernels

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
EXACT, SPLIT, HALF = kernels.GEMM_EXACT, kernels.GEMM_SPLIT_BF16, kernels.GEMM_SPLIT_F16


def hip():
    k = kernels.get()
    assert kernels.is_native()
    return k


@pytest.fixture(autouse=True)
def big_route(forced_big_route):
    """Every product of this file takes the 128 x 128 pipelined route (conftest.forced_big_route: cgc_gemm_tuning(11), as
    test_gemm_tail_split does): the route the mode applies to, at sizes a test can afford."""
    yield


def gen(shape, seed, kind, mn=-2):
    """``mn``: the axis of this operand that is an OUTPUT index (a row of op(A), a column of op(B)); the other matrix axis is k."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    if kind in ('wide', 'skewk') and len(shape) >= 2:
        # wide: every output row / column its own scale, 2^-30 .. 2^+30.  skewk: the scales run along K instead -- in both operands, so
        # the terms of every sum span 2^+-60 and one or two of them ARE the sum
        ax = (mn if kind == 'wide' else (-1 if mn == -2 else -2)) % len(shape)
        sh = [1] * len(shape)
        sh[ax] = shape[ax]
        x = x * torch.exp2(torch.randint(-30, 31, sh, generator=g).float())
    elif kind == 'tiny':                     # near the bottom of the exponent range: the lo plane is still a normal bf16
        x = x * 2.0 ** -100
    return x.to(DEV)


def mode_count(k, mode):
    """launches so far of the kernel that serves ``mode`` (0 for the exact one: it has no counter)"""
    return int(k.lib.cgc_gemm_split_count()) if mode == SPLIT else int(k.lib.cgc_gemm_half_count()) if mode == HALF else 0


def both_modes(run, want, mag, modes=None):
    """run() -> output tensor under k.gemm_mode; returns {mode: (max, rms)} of |out - want| / mag and asserts that the kernel of the mode
    really ran (and no other 16-bit kernel did)."""
    k = hip()
    res = {}
    for mode in (modes or (EXACT, SPLIT)):
        before = {m: mode_count(k, m) for m in (SPLIT, HALF)}
        k.gemm_mode = mode
        try:
            out = run()
        finally:
            k.gemm_mode = EXACT
        torch.cuda.synchronize()
        for m in (SPLIT, HALF):
            ran = mode_count(k, m) - before[m]
            assert (ran > 0) == (mode == m), (mode, m, ran)
        e = (out.double() - want).abs() / mag
        assert torch.isfinite(out).all()
        res[mode] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    return res


U = 2.0 ** -24          # unit roundoff of fp32 (round to nearest)


def skew_bars(K, mode):
    """Bars for the 'skewk' family, from each kernel's OWN accumulation structure instead of from the other kernel's measurement.

    Yardstick as everywhere in this file: e = |out - exact| / mag, mag = sum_k |a_k| |b_k| (+ |beta C| + |bias|).

    Exact kernel: v_mfma_f32_32x32x2_f32 is an fmaf chain -- n = K roundings (round to nearest), each of at most U x |partial sum|
    and |partial sum| <= mag: deterministically e <= K U; as a random walk of independent roundings (uniform in +-ulp/2: standard
    deviation ulp / sqrt(12) <= 2 U |partial| / sqrt(12)) rms(e) <= sqrt(K) U / sqrt(3).

    Split kernel, three terms:
      (a) the dropped pairs.  hi = RN_bf16(x), mid = RN_bf16(x - hi), lo = x - hi - mid (exact): |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, so
          |a_m b_l + a_l b_m + a_l b_l| <= (2^-24 + 2^-24 + 2^-32) |a||b| <= 3 U |a||b| per term of the sum: at most 3 U in e.
      (b) the accumulation: per k-tile of 16 and pair one v_mfma_f32_32x32x16_bf16 adds sixteen EXACT products to the accumulator.
          Model: its internal sum and the final add commit at most two roundings of <= U (|accumulator| + sum |products|) <= U mag each
          (the hardware's internal order is not documented; two roundings per instruction is the allowance): n = 2 x 6 x ceil(K / 16)
          roundings -- fewer than the exact chain's K for every K >= 48.
      (c) the epilogue (alpha, beta C, bias): a few roundings, common to both kernels, inside the +4.
    deterministically e <= (3 + n + 4) U; as a random walk rms(e) <= (sqrt(n + 4) / sqrt(3) + 3 / 2) U  (the dropped pairs do not
    average out by the same law: half their bound is allowed in full).

    Asserted: max(e) <= the deterministic bound AND max(e) <= 3.5 x the random-walk sigma + dropped (3.5 sigma of ~10^4 .. 10^6 outputs
    of a distribution with bounded support), rms(e) <= the random-walk rms.  Returns (rms bar, max bar, deterministic bound)."""
    if mode == EXACT:
        n = K + 4
        return (n ** 0.5 / 3 ** 0.5) * U, 3.5 * (n ** 0.5 / 3 ** 0.5) * U, n * U
    n = 2 * 6 * (-(-K // 16)) + 4
    return (n ** 0.5 / 3 ** 0.5 + 1.5) * U, (3.5 * n ** 0.5 / 3 ** 0.5 + 3.0) * U, (n + 3) * U


def check(res, what, outputs, K=None):
    """rms: 1.25 x the exact kernel's.  max: 1.25 x where it is a stable statistic (>= 10^6 outputs); on the small shapes (tens of
    thousands of outputs) the maximum of either kernel moves by tens of per cent with the seed: 2 x.
    The 'skewk' inputs (scales of 2^+-30 along K in BOTH operands: one or two terms of 2^+-60 ARE the sum, every partial sum is as large
    as mag) are where the two kernels differ by construction -- the fmaf chain rounds K times, the six bf16 pairs 6 ceil(K / 16) times
    through an instruction with its own internal order -- so neither is held to the other there: EACH is held to the analytic bound of
    its own accumulation structure (skew_bars: deterministic worst case and random-walk rms / 3.5 sigma, the dropped pairs' 3 U on
    top for the split kernel).  Round 5 used measured ratios here (1.5 x rms, 1.5 - 3 x max); the ratios are still printed.
    On every other input -- the ones the 1.25 x bars are for -- the measured ratios are 0.80 - 1.01."""
    (em, er), (sm, sr) = res[EXACT], res[SPLIT]
    line = '%s: exact max %.2e rms %.2e | split max %.2e rms %.2e  (ratios %.2f / %.2f)' % (what, em, er, sm, sr, sm / max(em, 1e-30), sr / max(er, 1e-30))
    skew = what.endswith('skewk')
    if skew:
        assert K is not None
        for mode, (m, r) in ((EXACT, (em, er)), (SPLIT, (sm, sr))):
            rbar, mbar, det = skew_bars(K, mode)
            line += ' | %s: rms / bar %.2f, max / 3.5-sigma bar %.2f, max / worst case %.3f' % ('exact' if mode == EXACT else 'split', r / rbar, m / mbar, m / det)
            assert r <= rbar and m <= mbar and m <= det, (what, mode, (m, r), (rbar, mbar, det))
    print(line)
    path = os.environ.get('CGC_SPLIT_ERROR_TABLE')
    if path:
        with open(path, 'a') as fh:
            fh.write(line + '\n')
    if skew:
        return
    big = outputs >= 1000000
    assert sr <= 1.25 * er + 2e-9, (what, res)
    assert sm <= (1.25 if big else 2.0) * em + 2e-9, (what, res)
    assert sm < 1e-6                                              # and absolutely: fp32-grade


KINDS = ['normal', 'wide', 'skewk', 'tiny']


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('M,N,K,tA,tB', [(3000, 1140, 1140, False, False), (2500, 1140, 1140, False, True), (1140, 1140, 5000, True, False),
                                         (257, 130, 170, False, False), (700, 300, 2052, False, True), (300, 260, 176, True, False)])
def test_split_gemm_flat(M, N, K, tA, tB, kind):
    """Flat products, alpha / beta / bias through the epilogue, operands as column windows of wider buffers (row strides 4 floats past
    the extent); K = 1140 has a partial last k-tile of 4."""
    k = hip()
    up4 = lambda v: (v + 3) // 4 * 4
    lda, ldb = up4(M if tA else K) + 4, up4(K if tB else N) + 4
    A = gen((K, lda) if tA else (M, lda), 1, kind, -1 if tA else -2)
    B = gen((N, ldb) if tB else (K, ldb), 2, 'normal' if kind == 'tiny' else kind, -2 if tB else -1)
    bias, C0 = gen((N,), 3, 'normal'), gen((M, N), 4, 'normal')
    a = (A[:, :M].t() if tA else A[:, :K]).double()
    b = (B[:, :K].t() if tB else B[:, :N]).double()
    scale = float((a.abs() @ b.abs()).mean())                    # beta * C0 and the bias at the scale of the product
    C0, bias = C0 * scale, bias * scale
    want = 0.5 * (a @ b) + 2.0 * C0.double() + bias.double()
    mag = 0.5 * (a.abs() @ b.abs()) + 2.0 * C0.double().abs() + bias.double().abs()

    def run():
        out = C0.clone()
        k.gemm(A, B, out, M, N, K, tA, tB, lda, ldb, N, 0.5, 2.0, bias)
        return out
    check(both_modes(run, want, mag), 'flat %s %s' % ((M, N, K, tA, tB), kind), M * N, K)


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('tB,xk,beta', [(False, 0, 0.0), (True, 20, 1.0), (False, 40, 0.0)])
def test_split_gemm_ragged_m_with_extra_segment(tB, xk, beta, kind):
    """Y_b = S_b op(G_b) (+ X_b op(H_b)) (+ Y_b): a B operand per graph, the rows of graph b (dP = S dA', dS += P dA'^T + X dX'^T)."""
    k = hip()
    counts = [700, 0, 513, 128, 77, 900, 250, 640]
    n, nmax, batch, N, K = sum(counts), max(counts), len(counts), 1140, 1140
    gptr = torch.tensor(np.cumsum([0] + counts), dtype=torch.int32, device=DEV)
    S = gen((n, K), 1, kind)
    G = gen((batch, N, K) if tB else (batch, K, N), 2, 'normal' if kind == 'tiny' else kind, -2 if tB else -1)
    X = gen((n, max(xk, 4)), 5, kind)
    H = gen((batch, N, max(xk, 4)) if tB else (batch, max(xk, 4), N), 6, 'normal' if kind == 'tiny' else kind, -2 if tB else -1)
    C0 = gen((n, N), 3, 'normal')
    gp = gptr.cpu().tolist()
    op = lambda t: t.double().t() if tB else t.double()
    parts, mags = [], []
    for b in range(batch):
        s, x = S[gp[b]:gp[b + 1]].double(), X[gp[b]:gp[b + 1], :xk].double()
        h = (H[b][:, :xk] if tB else H[b][:xk]).double()
        parts.append(s @ op(G[b]) + (x @ op(h) if xk else 0))
        mags.append(s.abs() @ op(G[b]).abs() + (x.abs() @ op(h).abs() if xk else 0))
    want, mag = torch.cat(parts), torch.cat(mags)
    C0 = C0 * float(mag.mean())
    want, mag = want + beta * C0.double(), mag + beta * C0.double().abs()

    def run():
        out = C0.clone()
        extra = [(X, H, X.shape[1], H.shape[2], xk, 0, H.shape[1] * H.shape[2])] if xk else ()
        k.gemm(S, G, out, 0, N, K, False, tB, K, G.shape[2], N, 1.0, beta, None, batch, 0, G.shape[1] * G.shape[2], 0, gptr, 1, nmax, n,
               extra=extra)
        return out
    check(both_modes(run, want, mag), 'ragged M tB=%s xk=%d %s' % (tB, xk, kind), n * N, K + xk)


@pytest.mark.parametrize('kind', KINDS)
def test_split_gemm_flat_with_two_extra_segments(kind):
    """Linear over cat[x1, x2, x3] (model/network.py:118-122): a long main segment and two short ones, K = 20 + 20 of which is all of a
    piece's pipeline (fewer k-tiles than the pipeline is deep)."""
    k = hip()
    M, N = 2100, 1140
    for K0 in (1140, 20):
        A, B = gen((M, K0), 1, kind), gen((K0, N), 2, 'normal' if kind == 'tiny' else kind, -1)
        X1, H1 = gen((M, 20), 3, kind), gen((20, N), 4, 'normal' if kind == 'tiny' else kind, -1)
        X2, H2 = gen((M, 24), 5, kind), gen((24, N), 6, 'normal' if kind == 'tiny' else kind, -1)
        bias = gen((N,), 7, 'normal') * (2.0 ** -100 if kind == 'tiny' else 1.0)
        want = A.double() @ B.double() + X1.double() @ H1.double() + X2.double() @ H2.double() + bias.double()
        mag = A.double().abs() @ B.double().abs() + X1.double().abs() @ H1.double().abs() + X2.double().abs() @ H2.double().abs() + bias.double().abs()

        def run():
            out = torch.full((M, N), float('nan'), device=DEV)
            k.gemm(A, B, out, M, N, K0, False, False, K0, N, N, 1.0, 0.0, bias, extra=[(X1, H1, 20, N, 20, 0, 0), (X2, H2, 24, N, 24, 0, 0)])
            return out
        check(both_modes(run, want, mag), 'cat K0=%d %s' % (K0, kind), M * N, K0 + 44)


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('counts', [[300, 0, 513, 128, 77, 900, 250, 640], [1800, 1900, 1750],
                                    [0, 300, 513, 128, 77, 900, 250, 640], [40, 300, 513, 128, 77, 900, 250, 640]])
def test_split_gemm_ragged_k(counts, kind):
    """out[b] = S_b^T P_b: the reduction runs over the rows of graph b (S^T (A S), S^T X).  The last two lists put an EMPTY graph and
    a graph of fewer k-tiles than tail-split pieces FIRST (row offset 0: a k range of no tiles must not load anything -- there is no
    row in front of the operand to clamp to; the kernel skips its prologue and stores zeros)."""
    k = hip()
    n, nmax, batch, C = sum(counts), max(counts), len(counts), 1140
    gptr = torch.tensor(np.cumsum([0] + counts), dtype=torch.int32, device=DEV)
    S, P = gen((n, C), 1, kind, -1), gen((n, C), 2, 'normal' if kind == 'tiny' else kind, -1)
    gp = gptr.cpu().tolist()
    want = torch.stack([S[gp[b]:gp[b + 1]].double().t() @ P[gp[b]:gp[b + 1]].double() for b in range(batch)])
    mag = torch.stack([S[gp[b]:gp[b + 1]].double().abs().t() @ P[gp[b]:gp[b + 1]].double().abs() for b in range(batch)])
    mag = mag.clamp_min(float(mag.max()) * 1e-30 + 1e-300)            # (an empty graph: 0 / 0)

    def run():
        out = torch.full((batch, C, C), float('nan'), device=DEV)
        k.gemm(S, P, out, C, C, 0, True, False, C, C, C, 1.0, 0.0, None, batch, 0, 0, C * C, gptr, 2, nmax, n)
        return out
    res = both_modes(run, want, mag)
    check(res, 'ragged K %s %s' % (counts[:3], kind), batch * C * C, nmax)


@pytest.mark.parametrize('kind', KINDS)
def test_split_gemm_uniform_k_chunks(kind):
    """ragged = 3 (the weight gradient X^T dY cut into row chunks, partial products + cgc_reduce_batch_sum)."""
    k = hip()
    M, N, Kd, chunk = 1140, 1140, 9000, 2080
    parts = -(-Kd // chunk)
    A, B = gen((Kd, M), 1, kind, -1), gen((Kd, N), 2, 'normal' if kind == 'tiny' else kind, -1)
    want = A.double().t() @ B.double()
    mag = A.double().abs().t() @ B.double().abs()

    def run():
        ws = torch.full((parts, M, N), float('nan'), device=DEV)
        k.gemm(A, B, ws, M, N, Kd, True, False, M, N, N, 1.0, 0.0, None, parts, 0, 0, M * N, None, 3, chunk, Kd)
        return ws.double().sum(0).float() if kind != 'tiny' else ws.double().sum(0)
    check(both_modes(run, want, mag), 'uniform chunks %s' % kind, M * N, Kd + parts)


@pytest.mark.parametrize('kind', ['normal', 'skewk'])
def test_split_gemm_dropped_pairs_are_bounded(kind):
    """Term (a) of skew_bars, checked on the operands themselves: the three planes of the kernel's split are re-formed with torch's own
    round-to-nearest-even bf16 conversion (the same rounding as v_cvt_pk_bf16_f32): hi + mid + lo == x EXACTLY, |mid| <= 2^-8 |x|,
    |lo| <= 2^-16 |x|, and the three pairs the kernel drops -- summed in float64 -- stay below 3 U of sum |a||b| on every output.  The
    kernel's result is then compared with the float64 product MINUS those pairs: what is left is its accumulation error alone.
    (Inputs in the normal range: at 2^-100 -- family 'tiny' -- the lo plane of the smaller elements falls below 2^-126 and the
    identity hi + mid + lo == x holds only to the denormal flush of whichever unit forms it; that family is held by the error bars of
    the other tests, as the kernel's header says.)"""
    k = hip()
    M, N, K = 700, 300, 1140
    A, B = gen((M, K), 11, kind), gen((K, N), 12, 'normal' if kind == 'tiny' else kind, -1)

    def planes(x):
        hi = x.bfloat16().float()
        r1 = x - hi
        mid = r1.bfloat16().float()
        lo = r1 - mid
        assert torch.equal(lo.bfloat16().float(), lo) and torch.equal((hi.double() + mid.double() + lo.double()).float(), x)
        ax = x.abs().double()
        assert bool((mid.abs().double() <= 2.0 ** -8 * ax).all()) and bool((lo.abs().double() <= 2.0 ** -16 * ax).all())
        return hi.double(), mid.double(), lo.double()
    (_, am, al), (_, bm, bl) = planes(A), planes(B)
    dropped = am @ bl + al @ bm + al @ bl
    want, mag = A.double() @ B.double(), A.double().abs() @ B.double().abs()
    assert bool((dropped.abs() <= 3 * U * mag).all()), float((dropped.abs() / mag).max())
    out = torch.empty(M, N, device=DEV)
    before = int(k.lib.cgc_gemm_split_count())
    k.gemm_mode = SPLIT
    try:
        k.gemm(A, B, out, M, N, K, False, False, K, N, N)
    finally:
        k.gemm_mode = EXACT
    torch.cuda.synchronize()
    assert int(k.lib.cgc_gemm_split_count()) == before + 1
    e_all = (out.double() - want).abs() / mag
    e_acc = (out.double() - (want - dropped)).abs() / mag
    rms = lambda e: float(e.pow(2).mean().sqrt())
    print('dropped pairs %s: max |dropped| / mag = %.2e (bound 3 U = %.2e), rms %.2e; kernel error vs the product: max %.2e rms %.2e; vs the product '
          'minus the dropped pairs: max %.2e rms %.2e' % (kind, float((dropped.abs() / mag).max()), 3 * U, rms(dropped / mag), float(e_all.max()), rms(e_all),
                                                           float(e_acc.max()), rms(e_acc)))
    rbar, mbar, det = skew_bars(K, SPLIT)
    assert float(e_acc.max()) <= mbar and rms(e_acc) <= rbar


@pytest.mark.parametrize('tA,tB', [(False, False), (False, True), (True, False)])
def test_split_gemm_strided_batch_and_tail_split_is_deterministic(tA, tB):
    """A strided batch whose tiles do not fill the chip (every tile cut along K into pieces + slab fix-up), twice: bitwise equal; and
    against the whole-tile launch (tail_split off): equal to rounding."""
    k = hip()
    batch, M, N, K = 3, 700, 600, 1333
    A = gen((batch, K, M) if tA else (batch, M, K), 1, 'normal')
    B = gen((batch, N, K) if tB else (batch, K, N), 2, 'normal')
    a = A.double().transpose(1, 2) if tA else A.double()
    b = B.double().transpose(1, 2) if tB else B.double()
    want, mag = torch.bmm(a, b), torch.bmm(a.abs(), b.abs())
    outs = []
    k.gemm_mode = SPLIT
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


def test_split_mode_leaves_other_routes_exact():
    """Products outside the 128 x 128 route (thin outputs, short reductions, operands unfit for 16-byte loads) run on the exact kernel
    whatever the mode says: bitwise the same result, no split launch."""
    k = hip()
    k.lib.cgc_gemm_tuning(0)                     # automatic routing here (the fixture restores its own setting afterwards)
    cases = [(5000, 40, 1140, False, False, 1140, 40), (3000, 1140, 100, False, True, 100, 100), (900, 700, 501, False, False, 501, 700)]
    for M, N, K, tA, tB, lda, ldb in cases:
        A, B = gen((M, lda), 1, 'normal'), gen((N if tB else K, ldb), 2, 'normal')
        outs = []
        for mode in (EXACT, SPLIT):
            before = int(k.lib.cgc_gemm_split_count())
            k.gemm_mode = mode
            try:
                out = torch.empty(M, N, device=DEV)
                k.gemm(A, B, out, M, N, K, tA, tB, lda, ldb, N)
            finally:
                k.gemm_mode = EXACT
            assert int(k.lib.cgc_gemm_split_count()) == before
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
