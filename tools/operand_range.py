#!/usr/bin/env python
"""Where the step's six dominant products sit relative to the input families of tests/test_split_gemm_gpu.py.

The split mode's error differs from the exact kernel's by construction on inputs whose magnitudes span many octaves ALONG K (family
'skewk': scales of 2^+-30 in both operands -- one or two terms ARE the sum).  This tool runs one training step of the benchmarked
batch (C3, per-operator path so that the products are visible from Python), taps the operands of every product on the 128 x 128
route and prints, per product:

  * spread along K of each operand: log2(max |x_k| / rms(x_k)) over K, median and maximum over the operand's output rows / columns;
  * top-2 share: for 8192 sampled outputs (i, j), the share of sum_k |a_ik| |b_kj| carried by its two largest terms (median / 99th
    percentile / max).  N(0, 1) inputs at K = 1140: ~0.007; family 'skewk': ~1.0.

GPU only.  Writes nothing; redirect stdout (tools/r06_measure.sh -> profiles/r06_operand_range_along_k.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgc_net_amd  # noqa: E402,F401
from cgc_net_amd import kernels, network  # noqa: E402
from cgc_net_amd.data import Batch, SyntheticCellGraphs  # noqa: E402

dev = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ds = SyntheticCellGraphs(B, 1800, 16, base_seed=0)
b = Batch.from_data_list([ds[i] for i in range(B)]).to(dev)
torch.manual_seed(0)
model = network.SoftPoolingGcnEncoder(11404, 16, 20, 20, True, True, 20, 3, 0.1, [50], concat=True, load_data_sparse=True,
                                      norm_adj=True, jk=True, drop_out=0.2).to(dev).train()
model.native = False
K = kernels.get()
orig = K.gemm
gen = torch.Generator(device='cpu').manual_seed(1)


def spread(x, kdim):
    """x [rows, cols] float64, K along ``kdim``: log2(max / rms) per output index."""
    a = x.abs()
    rms = a.pow(2).mean(kdim).sqrt().clamp_min(1e-300)
    s = torch.log2(a.amax(kdim).clamp_min(1e-300) / rms)
    live = a.amax(kdim) > 0
    s = s[live]
    return float(s.median()), float(s.max())


def report(tag, a, bm):
    """a [M, K], bm [K, N] (float64, one batch item / the flat product)."""
    M, Kd = a.shape
    N = bm.shape[1]
    i = torch.randint(0, M, (8192,), generator=gen).to(a.device)
    j = torch.randint(0, N, (8192,), generator=gen).to(a.device)
    t = a[i].abs() * bm[:, j].t().abs()                     # [8192, K]
    tot = t.sum(1).clamp_min(1e-300)
    top2 = t.topk(min(2, Kd), dim=1).values.sum(1) / tot
    sa, sb = spread(a, 1), spread(bm, 0)
    # mode CGC_GEMM_SPLIT_F16 (csrc/gemm_half.hip): the representation bound of the same outputs.  An element's error is at most
    # max(2^-22 |x|, 2^-24 / s) with s the scale of its panel (256 rows of op(A) / 128 columns of op(B); max |x| s in [2^14, 2^15)); the first term alone gives 2 x 2^-22 = 8 U of
    # sum |a||b| (U = 2^-24) on any input -- what is printed is how much the second term (elements more than 2^17 below their panel's
    # maximum) adds on THESE operands: 8.0 = nothing.
    def rep(x, axis, blk):
        """per panel (``blk`` consecutive output indices along ``axis``): scale and representation error of every element"""
        n = x.shape[axis]
        pad = (-n) % blk
        ax = x.abs()
        if axis == 1:
            ax = ax.t()
        mx = torch.nn.functional.pad(ax, (0, 0, 0, pad)).view(-1, blk, ax.shape[1]).amax(dim=(1, 2))          # [panels]
        sc = torch.where(mx > 0, torch.exp2(14 - torch.floor(torch.log2(mx.clamp_min(1e-300)))), torch.ones_like(mx))
        sc = sc.repeat_interleave(blk)[:n]
        sc = sc[:, None] if axis == 0 else sc[None, :]
        return torch.maximum(2.0 ** -22 * x.abs(), (2.0 ** -24 / sc).expand_as(x)), sc
    (da, s_a), (db, s_b) = rep(a, 0, 256), rep(bm, 1, 128)
    bnd = (da[i] * bm[:, j].t().abs() + a[i].abs() * db[:, j].t() + da[i] * db[:, j].t()).sum(1) / tot / 2.0 ** -24
    low_a = float((a.abs() * s_a < 2.0 ** -3).double().mean()), float((a.abs() * s_a < 2.0 ** -14).double().mean())
    low_b = float((bm.abs() * s_b < 2.0 ** -3).double().mean()), float((bm.abs() * s_b < 2.0 ** -14).double().mean())
    print('%-44s K = %-6d spread along K (log2 max / rms; median, max over outputs): A %.1f, %.1f   B %.1f, %.1f   top-2 share of sum |a||b|: '
          'median %.4f  p99 %.4f  max %.4f | fp16 mode: representation bound / (U sum |a||b|): median %.2f  p99 %.2f  max %.2f; elements with a '
          'subnormal l plane / subnormal h plane: A %.3f / %.5f  B %.3f / %.5f'
          % (tag, Kd, sa[0], sa[1], sb[0], sb[1], float(top2.median()), float(top2.quantile(0.99)), float(top2.max()),
             float(bnd.median()), float(bnd.quantile(0.99)), float(bnd.max()), low_a[0], low_a[1], low_b[0], low_b[1]))


def spy(A, Bm, C, M, N, Kd, tA, tB, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None, batch=1, sA=0, sB=0, sC=0, gptr=None,
        ragged=0, max_ragged=0, ragged_total=0, extra=()):
    rows = ragged_total if ragged == 1 else M
    kk = max_ragged if ragged >= 2 else Kd
    if N >= 1000 and rows >= 1000 and kk >= 1000:
        torch.cuda.synchronize()
        tag = '(M %d, N %d, K %d, %s%s, batch %d, ragged %d%s)' % (M, N, Kd, 'T' if tA else 'N', 'T' if tB else 'N', batch, ragged,
                                                                  ', + %s' % (tuple(e[4] for e in extra),) if extra else '')
        if ragged == 0 and batch == 1:
            a = (A[:Kd, :M].t() if tA else A[:M, :Kd]).double()
            bm = (Bm[:N, :Kd].t() if tB else Bm[:Kd, :N]).double()
        elif ragged == 1:                                    # rows of graph 0 against its own B
            g = gptr.cpu().tolist()
            a = A[g[0]:g[1], :Kd].double()
            b0 = Bm.view(-1)[:(N * ldb if tB else Kd * ldb)].view(-1, ldb)
            bm = (b0[:N, :Kd].t() if tB else b0[:Kd, :N]).double()
        elif ragged == 2:                                    # S_b^T P_b of graph 0: K = its rows
            g = gptr.cpu().tolist()
            a = A[g[0]:g[1], :M].t().double()
            bm = Bm[g[0]:g[1], :N].double()
        else:                                                # uniform row chunks: the first chunk
            a = A[:max_ragged, :M].t().double()
            bm = Bm[:max_ragged, :N].double()
        report(tag, a.contiguous(), bm.contiguous())
    return orig(A, Bm, C, M, N, Kd, tA, tB, lda, ldb, ldc, alpha, beta, bias, batch, sA, sB, sC, gptr, ragged, max_ragged, ragged_total, extra)


_, loss = model(b)
loss.backward()
model.zero_grad()
K.gemm = spy
print('# operands of the dominant products of one C3 training step (batch %d), family placement for tests/test_split_gemm_gpu.py' % B)
_, loss = model(b)
loss.backward()
torch.cuda.synchronize()
K.gemm = orig
# the synthetic families, same statistics
for kind, mk in (('normal', lambda s: torch.randn(*s, generator=gen)),
                 ('skewk', None)):
    a = torch.randn(2048, 1140, generator=gen)
    bm = torch.randn(1140, 1140, generator=gen)
    if kind == 'skewk':
        a = a * torch.exp2(torch.randint(-30, 31, (1, 1140), generator=gen).float())
        bm = bm * torch.exp2(torch.randint(-30, 31, (1140, 1), generator=gen).float())
    report('test family %r' % kind, a.double().to(dev), bm.double().to(dev))
