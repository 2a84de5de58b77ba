"""Autograd layer: each Function is a forward/backward pair scheduled over the kernel table
(kernels.py).  No arithmetic happens here -- only buffer allocation (torch caching allocator),
shape bookkeeping and the order in which C-ABI entry points are launched on the current stream.

Reference arithmetic being re-expressed:
  aggregate / linear_bias / l2_act_bn   DenseSAGEConv + act + BatchNorm of GNN_Module  (A4, A5; model/network.py:109-125)
  softmax_rows, diff_pool_*             _diff_pool                                     (A8; model/network.py:194-208)
  segment_max                           max readout incl. zero padding rows           (A9; model/network.py:264)
  rownorm_clamp, renorm_dense           clamp(min=1) mean divisor, _re_norm_adj at levels 2-3 (A4, A6)
"""
import torch
from torch.autograd import Function

from . import kernels
from .kernels import ACT_CODES


def K():
    return kernels.get()


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


_WIDE_MIN = 256      # (128 measured on the 180-cluster configuration: no gain)


def _wide(n, F, device):
    """[n, F] fp32 rows for kernel outputs.  Wide rows (F >= 256) get a row stride rounded up to 32 floats: every 128-byte
    line then belongs to ONE row, which the GEMM's k-contiguous tile loads (one line per 32-float segment instead of two)
    and the gather kernels reward with 5-15 % -- measured 109 -> 125 TFLOP/s on the NT contraction for 1140 vs 1152.
    Rows of 33..255 floats whose width is not a multiple of 4 (the level-2 cluster count 114) are padded to one: the GEMM's
    unguarded 16-byte loaders need 16-byte-aligned rows (csrc/gemm.hip)."""
    if F >= _WIDE_MIN and F % 32 != 0:
        ld = -(-F // 32) * 32
    elif F > 32 and F % 4 != 0:
        ld = -(-F // 4) * 4
    else:
        return torch.empty(n, F, dtype=torch.float32, device=device)
    # (an independent tensor on the padded storage, not a view: callers may modify it in place -- nn.ReLU(inplace=True) on a
    # custom Function's output that is a VIEW is refused by autograd)
    return _window(torch.empty(n, ld, dtype=torch.float32, device=device), 0, F)


def _pad4(b, r, c, device):
    """[b, r, c] fp32 for a batched product; large tensors whose rows are not a multiple of 4 floats get padded rows (see _wide)."""
    if c % 4 == 0 or c <= 32 or b * r * c < (1 << 20):
        return torch.empty(b, r, c, dtype=torch.float32, device=device)
    return torch.empty(b, r, -(-c // 4) * 4, dtype=torch.float32, device=device)[:, :, :c]


def _like3(t):
    """empty_like that keeps the row padding of a [b, r, c] tensor (torch.empty_like makes non-dense tensors contiguous)."""
    if t.is_contiguous():
        return torch.empty_like(t)
    b, r, c = t.shape
    return torch.empty(b, r, t.stride(1), dtype=t.dtype, device=t.device)[:, :, :c]


def _b3(t):
    """A [b, r, c] fp32 operand for _bgemm: as it is when its rows are contiguous and the batch is evenly strided (plain or with
    padded rows), else a contiguous copy."""
    if (t.dim() == 3 and t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) >= t.shape[2]
            and t.stride(0) == t.shape[1] * t.stride(1)):
        return t
    return _f32c(t)


def _window(buf, off, width):
    """Columns [off, off + width) of the 2-D buffer as an INDEPENDENT tensor on the same storage (not an autograd view of
    ``buf``): several custom Functions may each return a window of one buffer without tripping the view + in-place checks."""
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
        buf.untyped_storage(), buf.storage_offset() + off, (buf.shape[0], width), (buf.stride(0), 1))


def _rows_ld(t):
    """(tensor, ld) for a 2-D fp32 tensor whose rows are contiguous (e.g. a column slice of a wider one)."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.dtype == torch.float32:
        return t, t.stride(0)
    t = _f32c(t)
    return t, t.shape[-1]


# ----------------------------------------------------------------------------------------------
# split-K helper for the tall-skinny "weight gradient" contractions  out[Fa,Fb] = A[n,Fa]^T B[n,Fb]
# ----------------------------------------------------------------------------------------------
def _split_chunk(n, parts):
    """Rows per slice when n rows are cut into `parts` slices: a multiple of the GEMM's k-tile (32).  The slices are addressed by
    cgc_gemm_f32's ``ragged = 3`` mode (uniform row chunks): no offset tensor -- round 2 kept device tensors of split points in a
    bounded cache, which a captured hipGraph could outlive."""
    chunk = -(-n // parts)
    return -(-chunk // 32) * 32


_RESIDENT_BLOCKS = 512      # 256 CUs x 2 workgroups (the GEMM's ~70 KB of LDS admits two per CU)


def _split_parts(Fa, Fb, n, _cache={}):
    """How many row slices to cut the reduction into: fill whole rounds of resident workgroups, >= 512 rows per slice."""
    key = (Fa, Fb, n)
    if key not in _cache:
        tm = 32 if Fa <= 32 else 64 if Fa <= 64 else 128
        tn = 32 if Fb <= 32 else 64 if Fb <= 64 else 128
        tiles = (-(-Fa // tm)) * (-(-Fb // tn))
        best, best_score = 1, -1.0
        for parts in range(1, max(1, min(128, n // 512)) + 1):
            blocks = tiles * parts
            rounds = -(-blocks // _RESIDENT_BLOCKS)
            fill = blocks / float(rounds * _RESIDENT_BLOCKS)          # occupancy of the rounds it takes
            score = fill - 0.02 * rounds - (0.5 if blocks < 256 else 0.0)
            if score > best_score + 1e-9:
                best, best_score = parts, score
        if len(_cache) > 4096:       # n is a batch's node count: keep the memo bounded over a long run
            _cache.clear()
        _cache[key] = best
    return _cache[key]


def gemm_tn_rows(A, lda, Fa, B, ldb, Fb, n, out, ldc=None, beta=0.0):
    """out[Fa,Fb] (ld ldc) = beta*out + A[:n,:Fa]^T @ B[:n,:Fb]; rows split over workgroups, combined deterministically."""
    ldc = Fb if ldc is None else ldc
    parts = _split_parts(Fa, Fb, n)
    if parts == 1 or ldc != Fb:
        # (strided destinations take the direct path; they only occur for small slices)
        K().gemm(A, B, out, Fa, Fb, n, True, False, lda, ldb, ldc, 1.0, beta)
        return
    chunk = _split_chunk(n, parts)
    parts = -(-n // chunk)                                   # (slices that the rounding left empty are dropped)
    ws = torch.empty(parts, Fa * Fb, dtype=torch.float32, device=A.device)
    K().gemm(A, B, ws, Fa, Fb, n, True, False, lda, ldb, Fb, 1.0, 0.0, None,
             parts, 0, 0, Fa * Fb, None, 3, chunk, n)
    K().reduce_batch_sum(ws, out, parts, Fa * Fb, beta)


# ----------------------------------------------------------------------------------------------
# level-1 neighbour aggregation on the CSR (K1 narrow / K4 wide SpMM)
# ----------------------------------------------------------------------------------------------
class _Aggregate(Function):
    @staticmethod
    def forward(ctx, x, g, mean):
        x = _f32c(x)
        out = torch.empty_like(x)
        K().spmm(g.rowptr, g.col, None, g.val, None, g.inv_d if mean else None, x, out, g.n, x.shape[1],
                 g.gptr, g.B, g.nmax)
        ctx.g, ctx.mean = g, mean
        return out

    @staticmethod
    def backward(ctx, dy):
        g = ctx.g
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        # transpose aggregation: dx[j] = sum_i w_ij * inv_d[i] * dy[i]
        K().spmm(g.t_rowptr, g.t_col, None, g.t_val,
                 g.inv_d if ctx.mean else None, None, dy, dx, g.n, dy.shape[1], g.gptr, g.B, g.nmax)
        return dx, None, None


def aggregate(x, g, mean=True):
    """(A x) / clamp(rowsum A, 1)  (mean=True, DenseSAGEConv) or plain A x (mean=False: A S of _diff_pool, GIN)."""
    return _Aggregate.apply(x, g, mean)


class _SplitCols(Function):
    """(x[:, :w], x[:, w:]) as column windows of x.  Plain slicing would do the same forward; its autograd backward, however,
    builds each window's gradient as zeros + copy into a full-size tensor and then adds the two (5 kernels); here the two
    gradients are joined by one concatenation."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.w, ctx.shape = w, x.shape
        return x[:, :w], x[:, w:]

    @staticmethod
    def backward(ctx, ga, gb):
        n, tot = ctx.shape
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros(n, ctx.w)
        if gb is None:
            gb = ref.new_zeros(n, tot - ctx.w)
        return torch.cat([ga, gb], dim=1), None


def split_cols(x, w):
    return _SplitCols.apply(x, w)


# ----------------------------------------------------------------------------------------------
# y = x W (+ b)
# ----------------------------------------------------------------------------------------------
class _LinearBias(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, out_in_layout):
        (x, ldx), weight = _rows_ld(x), _f32c(weight)
        n, fin = x.shape
        fout = weight.shape[0] if out_in_layout else weight.shape[1]
        y = torch.empty(n, fout, dtype=torch.float32, device=x.device)
        K().gemm(x, weight, y, n, fout, fin, False, out_in_layout, ldx, weight.shape[1], fout, 1.0, 0.0, bias)
        ctx.save_for_backward(x, weight)
        ctx.out_in, ctx.has_bias = out_in_layout, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy, ld = _rows_ld(dy)
        n, fin = x.shape
        ldx = x.stride(0)
        fout = dy.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, fin, dtype=torch.float32, device=dy.device)
            # dx = dy W^T : with W [in,out] that is op(B)=W^T (transB); with W [out,in] it is plain W
            K().gemm(dy, weight, dx, n, fin, fout, False, not ctx.out_in, ld, weight.shape[1], fin)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            if ctx.out_in:      # dW[out,in] = dy^T x
                gemm_tn_rows(dy, ld, fout, x, ldx, fin, n, dw)
            else:               # dW[in,out] = x^T dy
                gemm_tn_rows(x, ldx, fin, dy, ld, fout, n, dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(fout, dtype=torch.float32, device=dy.device)
            K().colsum(dy, ld, n, fout, db)
        return dx, dw, db, None


def linear_bias(x, weight, bias=None, out_in_layout=False):
    return _LinearBias.apply(x, weight, bias, out_in_layout)


class _LinearCat(Function):
    """y = cat(xs, dim=1) @ W^T + b with W in nn.Linear layout [out, sum F_k]; the concatenation is never formed:
    the widest x_k is the main operand of one GEMM, the others ride along as extra K segments (model/network.py:118-122).
    Wide operands / results keep the padded row stride of ``_wide`` (strided views in, strided views out)."""

    @staticmethod
    def forward(ctx, weight, bias, softmax, *xs):
        xl = [_rows_ld(x) for x in xs]                    # (tensor, row stride)
        xs = [t for t, _ in xl]
        weight = _f32c(weight)
        n, fout, ftot = xs[0].shape[0], weight.shape[0], weight.shape[1]
        offs, o = [], 0
        for x in xs:
            offs.append(o)
            o += x.shape[1]
        assert o == ftot and len(xs) <= 3
        order = sorted(range(len(xs)), key=lambda i: -xs[i].shape[1])
        m = order[0]
        y = _wide(n, fout, weight.device)
        ldy = y.stride(0)
        if n >= 8 * fout:
            # tall products: the GEMM streams an n-contiguous B operand faster than a k-contiguous one, and transposing the
            # small weight costs microseconds; the copy gets the padded row stride as well
            wt = _wide(ftot, fout, weight.device)
            wt.copy_(weight.t())                                           # [sum F_k, out]
            ldw = wt.stride(0)
            extra = [(xs[i], wt[offs[i]:], xl[i][1], ldw, xs[i].shape[1], 0, 0) for i in order[1:]]
            K().gemm(xs[m], wt[offs[m]:], y, n, fout, xs[m].shape[1], False, False, xl[m][1], ldw, ldy,
                     1.0, 0.0, bias, extra=extra)
        else:
            extra = [(xs[i], weight[:, offs[i]:], xl[i][1], ftot, xs[i].shape[1], 0, 0) for i in order[1:]]
            K().gemm(xs[m], weight[:, offs[m]:], y, n, fout, xs[m].shape[1], False, True, xl[m][1], ftot, ldy,
                     1.0, 0.0, bias, extra=extra)
        if softmax:                      # row softmax of the assignment logits, in place (model/network.py:200)
            K().softmax_fwd(y, n, fout, y, ldy)
            ctx.save_for_backward(weight, y, *xs)
        else:
            ctx.save_for_backward(weight, *xs)
        ctx.offs, ctx.has_bias, ctx.softmax = offs, bias is not None, softmax
        return y

    @staticmethod
    def backward(ctx, dy):
        weight = ctx.saved_tensors[0]
        dw = db = None
        if ctx.softmax:
            s_out, xs = ctx.saved_tensors[1], ctx.saved_tensors[2:]
            n_, c_ = s_out.shape
            lds = s_out.stride(0)
            ds, ldd = _rows_ld(dy)
            if ldd != lds:                                 # foreign gradient layout: bring it to the activation's
                t = _wide(n_, c_, s_out.device)
                t.copy_(ds)
                ds = t
            dy = _wide(n_, c_, s_out.device)
            if ctx.has_bias and ctx.needs_input_grad[1]:
                db = torch.empty(c_, dtype=torch.float32, device=ds.device)
            K().softmax_bwd(s_out, ds, n_, c_, dy, db, lds)               # + column sums = bias gradient
            ld = dy.stride(0)
        else:
            xs = ctx.saved_tensors[1:]
            dy, ld = _rows_ld(dy)
        n, fout, ftot = dy.shape[0], weight.shape[0], weight.shape[1]
        dxs = [None] * len(xs)
        if any(ctx.needs_input_grad[3:]):
            # d x_k = dy W[:, slice_k]: one product per piece.  A single product over all sum F_k = 1180 columns would run ten
            # 128-wide column tiles for 9.2 tiles of work (8 % of a 155 GFLOP contraction) and leave d x_3 with an unaligned
            # 4720-byte row stride; per piece the wide one has exactly nine tiles and its rows get the padded stride, and the
            # narrow one is a skinny product that re-reads dy once (~60 us).
            for i, (o, x, need) in enumerate(zip(ctx.offs, xs, ctx.needs_input_grad[3:])):
                if not need:
                    continue
                f = x.shape[1]
                if (o * 4) % 16 != 0:                                    # slice start not 16-byte aligned: one joint product
                    dcat = torch.empty(n, ftot, dtype=torch.float32, device=dy.device)
                    K().gemm(dy, weight, dcat, n, ftot, fout, False, False, ld, ftot, ftot)
                    dxs = [dcat[:, oo:oo + xx.shape[1]] if nn_ else None
                           for oo, xx, nn_ in zip(ctx.offs, xs, ctx.needs_input_grad[3:])]
                    break
                dxk = _wide(n, f, dy.device)
                K().gemm(dy, weight[:, o:], dxk, n, f, fout, False, False, ld, ftot, dxk.stride(0))
                dxs[i] = dxk
        if ctx.needs_input_grad[0]:
            dw = torch.empty_like(weight)
            for o, x in zip(ctx.offs, xs):                                          # dW[:, slice_k] = dy^T x_k
                f = x.shape[1]
                tmp = torch.empty(fout, f, dtype=torch.float32, device=dy.device)
                gemm_tn_rows(dy, ld, fout, x, x.stride(0), f, n, tmp)
                dw[:, o:o + f].copy_(tmp)
        if ctx.has_bias and ctx.needs_input_grad[1] and db is None:
            db = torch.empty(fout, dtype=torch.float32, device=dy.device)
            K().colsum(dy, ld, n, fout, db)
        return (dw, db, None) + tuple(dxs)


def linear_cat(xs, weight, bias=None, softmax=False):
    """nn.Linear applied to torch.cat(xs, dim=1) (at most 3 pieces) without forming the concatenation;
    ``softmax=True`` appends the row softmax (the assignment matrix of _diff_pool) in the same autograd node."""
    return _LinearCat.apply(weight, bias, softmax, *xs)


# ----------------------------------------------------------------------------------------------
# row L2-normalise -> activation -> BatchNorm (statistics over `count` rows, zero padding included)
# ----------------------------------------------------------------------------------------------
class _L2ActBN(Function):
    @staticmethod
    def forward(ctx, h, gamma, beta, running_mean, running_var, count, act, normalize, bn_mode, eps, momentum, nbt=None):
        h = _f32c(h)
        n, F = h.shape
        dev = h.device
        hn = torch.empty_like(h)
        rinv = torch.empty(n, dtype=torch.float32, device=dev)
        mean = istd = None
        if bn_mode == 2:
            mean = torch.empty(F, dtype=torch.float32, device=dev)
            istd = torch.empty(F, dtype=torch.float32, device=dev)
            K().l2norm_act_bn(h, n, F, normalize, act, hn, rinv, float(count), eps, momentum, running_mean, running_var, nbt,
                              mean, istd)
        else:
            K().l2norm_act_stats(h, n, F, normalize, act, hn, rinv, None)
            if bn_mode == 1:
                mean, istd = running_mean, torch.rsqrt(running_var + eps)
        y = torch.empty_like(h)
        K().bn_act_apply(hn, n, F, act, mean, istd, gamma, beta, y, F)
        ctx.save_for_backward(hn, rinv, mean, istd, gamma)
        ctx.cfg = (act, normalize, bn_mode, float(count))
        return y

    @staticmethod
    def backward(ctx, dy):
        hn, rinv, mean, istd, gamma = ctx.saved_tensors
        act, normalize, bn_mode, count = ctx.cfg
        n, F = hn.shape
        dy, ld = _rows_ld(dy)
        sums = dgamma = dbeta = None
        if bn_mode != 0:
            sums = torch.empty(2, F, dtype=torch.float32, device=hn.device)
            K().bn_bwd_reduce(dy, ld, hn, n, F, act, mean, istd, sums)
            dbeta, dgamma = sums[0], sums[1]
        dh = torch.empty_like(hn)
        K().bn_act_l2_bwd(dy, ld, hn, rinv, n, F, act, normalize, bn_mode, mean, istd, gamma, sums, count, dh)
        return dh, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def _bn_momentum(bn, count_batch):
    """nn.BatchNorm1d's bookkeeping: num_batches_tracked += 1 per training forward; momentum=None means the cumulative
    moving average 1/num_batches_tracked.  Returns (momentum, counter tensor for the KERNEL to increment or None)."""
    nbt = bn.num_batches_tracked if count_batch else None
    if bn.momentum is not None:
        return bn.momentum, nbt                     # the counter rides along with the statistics kernel
    if nbt is not None:
        nbt.add_(1)                                  # cumulative average: the value is needed on the host first
    return 1.0 / float(max(int(bn.num_batches_tracked), 1)), None


def l2_act_bn(h, bn, count, act='relu', normalize=True, training=True):
    """BN(act(l2norm(h))) with BatchNorm statistics over ``count`` rows (rows beyond h's are zeros).

    ``bn`` is an nn.BatchNorm1d (parameter/buffer holder) or None.  In training mode the running
    statistics and num_batches_tracked are updated as nn.BatchNorm1d does.
    """
    code = ACT_CODES[act]
    if bn is None:
        return _L2ActBN.apply(h, None, None, None, None, count, code, normalize, 0, 0.0, 0.0)
    use_batch = training or bn.running_mean is None
    momentum, nbt = _bn_momentum(bn, use_batch and training)
    rm, rv = (bn.running_mean, bn.running_var) if (training and bn.track_running_stats) else (None, None)
    if use_batch:
        return _L2ActBN.apply(h, bn.weight, bn.bias, rm, rv, count, code, normalize, 2, bn.eps, momentum, nbt)
    return _L2ActBN.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, count, code, normalize, 1,
                          bn.eps, 0.0)


class _SageProject(Function):
    """BN(act(l2norm(agg @ W + b))): the tail of one SAGE convolution as ONE autograd node.  Forward = the launches of
    linear_bias + l2_act_bn; backward computes d(agg W + b) with the fused BN/act/l2 kernel, which also emits its column
    sums (= db), then dW (row-split GEMM) and d agg."""

    @staticmethod
    def forward(ctx, agg, weight, bias, gamma, beta, running_mean, running_var, count, act, normalize, bn_mode, eps, momentum,
                nbt=None, out=None):
        (agg, lda), weight = _rows_ld(agg), _f32c(weight)      # agg may be a column slice of a paired aggregation
        n, fin = agg.shape
        F = weight.shape[1]
        dev = agg.device
        h = torch.empty(n, F, dtype=torch.float32, device=dev)
        rinv = torch.empty(n, dtype=torch.float32, device=dev)
        mean = istd = None
        if bn_mode == 2:
            mean = torch.empty(F, dtype=torch.float32, device=dev)
            istd = torch.empty(F, dtype=torch.float32, device=dev)
        # narrow input (hidden width): projection, L2 normalisation and BatchNorm statistics in ONE kernel that writes only hn --
        # for the wide output of the assignment block's last convolution ([Ntot, 20] -> [Ntot, 1140]) and for the narrow layers;
        # everything else: MFMA GEMM + row kernel
        fused = (F >= _WIDE_MIN or F <= 32) and fin <= 32 and K().sage_wide_fwd(
            agg, lda, weight, bias, n, fin, F, normalize, act, h, rinv, bn_mode == 2, float(count), eps, momentum,
            running_mean, running_var, nbt, mean, istd)
        if not fused:
            K().gemm(agg, weight, h, n, F, fin, False, False, lda, F, F, 1.0, 0.0, bias)
            if bn_mode == 2:
                K().l2norm_act_bn(h, n, F, normalize, act, h, rinv, float(count), eps, momentum, running_mean, running_var, nbt,
                                  mean, istd)                                      # in place: h becomes hn
            else:
                K().l2norm_act_stats(h, n, F, normalize, act, h, rinv, None)
        if bn_mode == 1:
            mean, istd = running_mean, torch.rsqrt(running_var + eps)
        if out is not None:                               # (buffer, column offset): the result lands in a window of a shared buffer
            y = _window(out[0], out[1], F)                # (the two blocks of a level write side by side: no concatenation later)
        else:
            y = _wide(n, F, dev)                          # the 1140-wide layer output is the A operand of the assignment Linear
        K().bn_act_apply(h, n, F, act, mean, istd, gamma, beta, y, y.stride(0))
        ctx.save_for_backward(agg, weight, h, rinv, mean, istd, gamma)
        ctx.cfg = (act, normalize, bn_mode, float(count), bias is not None, lda)
        return y

    @staticmethod
    def backward(ctx, dy):
        agg, weight, hn, rinv, mean, istd, gamma = ctx.saved_tensors
        act, normalize, bn_mode, count, has_bias, lda = ctx.cfg
        n, F = hn.shape
        fin = agg.shape[1]
        dy, ld = _rows_ld(dy)
        dev = hn.device
        sums = dgamma = dbeta = None
        if bn_mode != 0:
            sums = torch.empty(2, F, dtype=torch.float32, device=dev)
            K().bn_bwd_reduce(dy, ld, hn, n, F, act, mean, istd, sums)
            dbeta, dgamma = sums[0], sums[1]
        if F <= 32 and fin <= 32 and ctx.needs_input_grad[1]:
            # narrow layer: BN / activation / L2-norm backward and its three consumers (d agg, d W, d b) in ONE kernel; dh is never
            # written, the 20 x 20 weight gradient is accumulated on the matrix cores instead of a split-K GEMM over 57.7 k rows
            dagg = torch.empty(n, fin, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
            dwdb = torch.empty(fin * F + F, dtype=torch.float32, device=dev)
            if K().sage_narrow_bwd(dy, ld, hn, rinv, n, F, act, normalize, bn_mode, mean, istd, gamma, sums, count, agg, lda, fin,
                                   weight, dagg, dwdb):
                dw = dwdb[:fin * F].view(fin, F)
                db = dwdb[fin * F:] if has_bias else None
                return dagg, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None
        dh = torch.empty_like(hn)
        db = torch.empty(F, dtype=torch.float32, device=dev) if has_bias else None
        K().bn_act_l2_bwd(dy, ld, hn, rinv, n, F, act, normalize, bn_mode, mean, istd, gamma, sums, count, dh, db)
        dagg = dw = None
        if ctx.needs_input_grad[0]:
            dagg = torch.empty(n, fin, dtype=torch.float32, device=dev)
            K().gemm(dh, weight, dagg, n, fin, F, False, True, F, F, fin)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            gemm_tn_rows(agg, lda, fin, dh, F, F, n, dw)
        return dagg, dw, db, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def sage_project(agg, weight, bias, bn, count, act='relu', normalize=True, training=True, out=None):
    """One node for ``l2_act_bn(linear_bias(agg, weight, bias), bn, ...)`` (weight in PyG layout [in, out]).
    ``out = (buffer [rows, >= off + F], off)``: write the result into that column window instead of a fresh tensor."""
    code = ACT_CODES[act]
    if bn is None:
        return _SageProject.apply(agg, weight, bias, None, None, None, None, count, code, normalize, 0, 0.0, 0.0, None, out)
    use_batch = training or bn.running_mean is None
    momentum, nbt = _bn_momentum(bn, use_batch and training)
    rm, rv = (bn.running_mean, bn.running_var) if (training and bn.track_running_stats) else (None, None)
    if use_batch:
        return _SageProject.apply(agg, weight, bias, bn.weight, bn.bias, rm, rv, count, code, normalize, 2, bn.eps, momentum,
                                  nbt, out)
    return _SageProject.apply(agg, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, count, code,
                              normalize, 1, bn.eps, 0.0, None, out)


class _JoinCols(Function):
    """The [rows, wa + wb] buffer whose two column windows ARE a and b (both were written there by sage_project(out=...)): the
    concatenation without a copy.  Backward hands each producer its window of the gradient."""

    @staticmethod
    def forward(ctx, a, b, holder):
        buf = holder[0]
        assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf[:, a.shape[1]:].data_ptr() and a.stride(0) == buf.stride(0)
        ctx.wa = a.shape[1]
        return _window(buf, 0, buf.shape[1])

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.wa], g[:, ctx.wa:], None


def join_cols(a, b, buf):
    return _JoinCols.apply(a, b, (buf,))


# ----------------------------------------------------------------------------------------------
# assignment softmax, max readout
# ----------------------------------------------------------------------------------------------
class _SoftmaxRows(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        s = torch.empty_like(x)
        K().softmax_fwd(x, x.shape[0], x.shape[1], s)
        ctx.save_for_backward(s)
        return s

    @staticmethod
    def backward(ctx, ds):
        s, = ctx.saved_tensors
        ds = _f32c(ds)
        dx = torch.empty_like(s)
        K().softmax_bwd(s, ds, s.shape[0], s.shape[1], dx)
        return dx


def softmax_rows(x):
    return _SoftmaxRows.apply(x)


class _SegmentMax(Function):
    @staticmethod
    def forward(ctx, x, gptr, B, nmax):
        x = _f32c(x)
        D = x.shape[1]
        out = torch.empty(B, D, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, D, dtype=torch.int32, device=x.device)
        K().segment_max_fwd(x, gptr, B, D, nmax, out, arg)
        ctx.save_for_backward(arg, gptr)
        ctx.n, ctx.nmax = x.shape[0], nmax
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, gptr = ctx.saved_tensors
        B, D = arg.shape
        dx = torch.empty(ctx.n, D, dtype=torch.float32, device=dout.device)
        K().segment_max_bwd_full(_f32c(dout), arg, gptr, B, D, ctx.nmax, dx)       # writes every element: no zero fill
        return dx, None, None, None


def segment_max(x, gptr, B, nmax):
    """[n, D] -> [B, D]: per-graph max over nodes; graphs shorter than nmax also see the zero padding rows."""
    return _SegmentMax.apply(x, gptr, B, nmax)


# ----------------------------------------------------------------------------------------------
# jumping-knowledge attention (DenseJK): fused bi-LSTM + attention, one thread per node
# ----------------------------------------------------------------------------------------------
class _DenseJK(Function):
    @staticmethod
    def forward(ctx, xs, w_att, b_att, *lstm):
        xs = _f32c(xs)
        n = xs.shape[0]
        C = xs.shape[1] // 3
        H = 3 * C // 2
        npad = -(-max(n, 1) // 1024) * 1024            # row padding of the transposed buffers: K slices of 768 divide 3*npad
        dev = xs.device
        lstm = [_f32c(p) for p in lstm]
        w_att, b_att = _f32c(w_att), _f32c(b_att)
        out = torch.empty(n, C, dtype=torch.float32, device=dev)
        HS = torch.empty(6 * H, npad, dtype=torch.float32, device=dev)
        CS = torch.empty(6 * H, npad, dtype=torch.float32, device=dev)
        K().jk_fwd(xs, n, npad, C, lstm, w_att, b_att, out, HS, CS)
        ctx.save_for_backward(xs, w_att, b_att, HS, CS, *lstm)
        ctx.dims = (n, npad, C, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        xs, w_att, b_att, HS, CS = ctx.saved_tensors[:5]
        lstm = list(ctx.saved_tensors[5:])
        n, npad, C, H = ctx.dims
        dev = xs.device
        ng, ni, ktot = 4 * H + 1, C + 2 * H + 1, 3 * npad
        dxs = torch.empty_like(xs)
        # parameter gradients: G_d = (gate gradients | score gradient) x (x_t | h_{t-1} | 1 | h_t)^T, accumulated inside the
        # backward kernel on the matrix cores (staged through transposed buffers + a GEMM only on the unaligned fallback)
        G = torch.empty(2, ng, ni, dtype=torch.float32, device=dev)
        K().jk_bwd_params(xs, _f32c(dout), n, npad, C, lstm, w_att, b_att, HS, CS, dxs, G)
        # one unpack kernel -> contiguous gradients (strided windows of G would each be cloned by AccumulateGrad)
        g = K().jk_unpack_param_grads(G, C)
        dw_att, db_att = g[8].reshape(w_att.shape), g[9].reshape(b_att.shape)
        return (dxs, dw_att, db_att) + tuple(g[:8])


def dense_jk(xs, lstm_module, att_module):
    """DenseJK on flat rows: xs [rows, 3C] -> [rows, C] through the fused kernels (torch.nn.LSTM / nn.Linear hold the
    parameters; their layouts are consumed as they are)."""
    p = lstm_module
    lstm = (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0,
            p.weight_ih_l0_reverse, p.weight_hh_l0_reverse, p.bias_ih_l0_reverse, p.bias_hh_l0_reverse)
    return _DenseJK.apply(xs, att_module.weight, att_module.bias, *lstm)


# ----------------------------------------------------------------------------------------------
# DiffPool on the sparse level:  X' = S^T X,  A' = S^T (A S)   per graph
# ----------------------------------------------------------------------------------------------
class _DiffPoolSparse(Function):
    @staticmethod
    def forward(ctx, embed, s, g):
        embed = _f32c(embed)
        s, ld = _rows_ld(s)                                       # assignment rows keep their (padded) stride
        n, dx = embed.shape
        c = s.shape[1]
        dev = s.device
        if ld == c:
            p, pad = torch.empty_like(s), None                    # P = A S   (K4, wide SpMM)
        else:
            p, pad = _wide(n, c, dev), ld
            if p.stride(0) != ld:                                 # foreign stride: fall back to dense rows
                s, ld, pad = s.contiguous(), c, None
                p = torch.empty_like(s)
        K().spmm(g.rowptr, g.col, None, g.val, None, None, s, p, n, c, g.gptr, g.B, g.nmax, 1 | (4 if g.spatial else 0), pad, g.gorder)   # S: fresh from the softmax
        xo = torch.empty(g.B, c, dx, dtype=torch.float32, device=dev)
        ao = torch.empty(g.B, c, c, dtype=torch.float32, device=dev)
        # ragged-K, transposed-A contractions: per graph  [c x N_b] . [N_b x (dx | c)]
        K().gemm(s, embed, xo, c, dx, 0, True, False, ld, dx, dx, 1.0, 0.0, None, g.B, 0, 0, c * dx, g.gptr, 2, g.nmax, n)
        K().gemm(s, p, ao, c, c, 0, True, False, ld, ld, c, 1.0, 0.0, None, g.B, 0, 0, c * c, g.gptr, 2, g.nmax, n)
        ctx.save_for_backward(embed, s, p)
        ctx.g, ctx.ld = g, ld
        return xo, ao

    @staticmethod
    def backward(ctx, dxo, dao):
        embed, s, p = ctx.saved_tensors
        g, ld = ctx.g, ctx.ld
        n, dx = embed.shape
        c = s.shape[1]
        dev = s.device
        pad = None if ld == c else ld
        dxo, dao = _f32c(dxo), _f32c(dao)

        def rows():
            return torch.empty(n, c, dtype=torch.float32, device=dev) if pad is None else _wide(n, c, dev)
        # dP = S dA'   (ragged-M, NN)
        dp = rows()
        K().gemm(s, dao, dp, 0, c, c, False, False, ld, c, ld, 1.0, 0.0, None, g.B, 0, c * c, 0, g.gptr, 1, g.nmax, n)
        # dS = A^T dP  (transpose SpMM) + P dA'^T + X dX'^T
        ds = rows()
        K().spmm(g.t_rowptr, g.t_col, None, g.t_val, None, None, dp, ds, n, c,
                 g.gptr, g.B, g.nmax, 2 | (4 if g.spatial else 0), pad, g.gorder)                                              # dP: fresh from the gemm
        # ... both products in one launch: [P | X] [dA' | dX']^T, the K = dx segment rides on the K = c product
        K().gemm(p, dao, ds, 0, c, c, False, True, ld, c, ld, 1.0, 1.0, None, g.B, 0, c * c, 0, g.gptr, 1, g.nmax, n,
                 extra=[(embed, dxo, dx, dx, dx, 0, c * dx)])
        # dX = S dX'
        de = torch.empty_like(embed)
        K().gemm(s, dxo, de, 0, dx, c, False, False, ld, dx, dx, 1.0, 0.0, None, g.B, 0, c * dx, 0, g.gptr, 1, g.nmax, n)
        return de, ds, None


def diff_pool_sparse(embed, s, g):
    return _DiffPoolSparse.apply(embed, s, g)


# ----------------------------------------------------------------------------------------------
# strided-batched dense matmul C_b = op(A_b) op(B_b)  (levels 2-3: A~ x, S^T X, A S, S^T (A S))
# ----------------------------------------------------------------------------------------------
def _bgemm(A, B, C, tA, tB, beta=0.0):
    """A, B, C: [batch, r, c] tensors with contiguous (possibly padded) rows, see _b3; computes C = op(A) op(B) (+ beta C)."""
    batch = C.shape[0]
    M, N = C.shape[1], C.shape[2]
    Kd = A.shape[1] if tA else A.shape[2]
    lda, ldb, ldc = A.stride(1), B.stride(1), C.stride(1)
    if tA and not tB and M <= 128 and N <= 128 and Kd >= 512 and batch * 2 < 256 and ldc == N:
        # small outputs reduced over a long axis (S2^T P2, S2^T X at level 2): too few tiles to fill the chip -> cut every
        # batch's reduction into slices (ragged-K over the flattened rows), combine deterministically
        parts = max(2, min(Kd // 128, -(-384 // (batch * (2 if N > 64 else 1)))))
        step = _split_chunk(Kd, parts)
        parts = -(-Kd // step)
        ws = torch.empty(batch * parts, M * N, dtype=torch.float32, device=A.device)
        K().gemm(A, B, ws, M, N, Kd, True, False, lda, ldb, N, 1.0, 0.0, None, batch * parts, 0, 0, M * N,
                 None, 3, step, batch * Kd)
        K().reduce_batched(ws, C, batch, parts, M * N, beta)
        return
    K().gemm(A, B, C, M, N, Kd, tA, tB, lda, ldb, ldc, 1.0, beta, None, batch, A.stride(0), B.stride(0), C.stride(0))


class SharedGrad(object):
    """Deferred, batched gradient of ONE tensor A that feeds several ``C_i = A @ B_i`` nodes (the row-normalised adjacency
    of a dense level feeds every convolution of both blocks).  dA = sum_i dC_i B_i^T is a sum of short (K = 20..60)
    rank-k updates, each of which would write -- and, accumulating, re-read -- the full [B, C, C] tensor.  Instead every
    node only deposits (dC_i, B_i); the node that arrives last concatenates them along the feature axis and issues ONE
    product [dC_1 | dC_2 | ...] [B_1 | B_2 | ...]^T, so the [B, C, C] gradient is written once.  The other nodes report no
    gradient; the result does not depend on the order in which autograd runs the nodes.  Every registered node must take
    part in the backward pass (true for the encoder: all aggregations reach the loss)."""

    def __init__(self):
        self.uses = 0
        self.pending = 0
        self.parts = []

    def register(self):
        self.uses += 1
        self.pending = self.uses

    def contribute(self, like, dC, B):
        self.parts.append((dC, B))
        self.pending -= 1
        if self.pending > 0:
            return None
        parts, self.parts, self.pending = self.parts, [], self.uses
        g = parts[0][0] if len(parts) == 1 else torch.cat([p[0] for p in parts], dim=2)
        x = parts[0][1] if len(parts) == 1 else torch.cat([p[1] for p in parts], dim=2)
        dA = torch.empty_like(like)
        _bgemm(g, x, dA, False, True)          # dA = [dC_1|dC_2|..] [B_1|B_2|..]^T
        return dA


class _BMatmul(Function):
    @staticmethod
    def forward(ctx, A, B, tA, tB, shared=None):
        A, B = _b3(A), _b3(B)
        ctx.shared = shared
        if shared is not None:
            shared.register()
        M = A.shape[2] if tA else A.shape[1]
        N = B.shape[1] if tB else B.shape[2]
        C = _pad4(A.shape[0], M, N, A.device)
        _bgemm(A, B, C, tA, tB)
        ctx.save_for_backward(A, B)
        ctx.t = (tA, tB)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        tA, tB = ctx.t
        dC = _b3(dC)
        dA = dB = None
        if ctx.needs_input_grad[0]:
            if ctx.shared is not None:
                dA = ctx.shared.contribute(A, dC, B)          # plain A @ B nodes only (asserted in bmatmul)
            else:
                dA = _like3(A)
                if not tA:
                    _bgemm(dC, B, dA, False, not tB)          # dA = dC op(B)^T
                else:
                    _bgemm(B, dC, dA, tB, True)               # dA = op(B) dC^T
        if ctx.needs_input_grad[1]:
            dB = _like3(B)
            if not tB:
                _bgemm(A, dC, dB, not tA, False)        # dB = op(A)^T dC
            else:
                _bgemm(dC, A, dB, True, tA)             # dB = dC^T op(A)
        return dA, dB, None, None, None


def bmatmul(A, B, tA=False, tB=False, shared=None):
    """``shared``: a SharedGrad that every bmatmul using the same A passes (optional; see SharedGrad)."""
    assert not (tA and tB)
    assert shared is None or not (tA or tB)
    return _BMatmul.apply(A, B, tA, tB, shared)


def diff_pool_dense(embed, adj, s):
    """_diff_pool on dense tensors (levels >= 2): (S^T X, S^T (A S))."""
    return bmatmul(s, embed, tA=True), bmatmul(s, bmatmul(adj, s), tA=True)


# ----------------------------------------------------------------------------------------------
# dense adjacency transforms with gradient (levels 2-3)
# ----------------------------------------------------------------------------------------------
class _RowNormClamp(Function):
    @staticmethod
    def forward(ctx, A):
        A = _f32c(A)
        b, c, _ = A.shape
        out = torch.empty_like(A)
        invd = torch.empty(b * c, dtype=torch.float32, device=A.device)
        ge1 = torch.empty(b * c, dtype=torch.float32, device=A.device)
        K().dense_rownorm_fwd(A, b * c, c, out, invd, ge1)
        ctx.save_for_backward(out, invd, ge1)
        return out

    @staticmethod
    def backward(ctx, dOut):
        out, invd, ge1 = ctx.saved_tensors
        b, c, _ = out.shape
        dA = torch.empty_like(out)
        K().dense_rownorm_bwd(_f32c(dOut), out, invd, ge1, b * c, c, dA)
        return dA


def rownorm_clamp(A):
    """A / clamp(rowsum(A), min=1): DenseSAGEConv's mean divisor folded into the adjacency, computed once per level."""
    return _RowNormClamp.apply(A)


class _ReNormDense(Function):
    @staticmethod
    def forward(ctx, A, p):
        A = _f32c(A)
        b, c, _ = A.shape
        out = torch.empty_like(A)
        K().dense_renorm_fwd(A, b * c, c, p, out)
        ctx.save_for_backward(A)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, dOut):
        A, = ctx.saved_tensors
        b, c, _ = A.shape
        dA = torch.empty_like(A)
        K().dense_renorm_bwd(A, _f32c(dOut), b * c, c, ctx.p, dA)
        return dA, None


def renorm_dense(A, p):
    """_re_norm_adj (model/network.py:183-191) on a dense [B,C,C] adjacency that requires grad."""
    return _ReNormDense.apply(A, float(p))


class _AdjPrep(Function):
    """(A~, A~/clamp(rowsum A~, 1)) of a dense level in one pass, A~ = _re_norm_adj(A, p) or A itself (p None).  Both results
    are outputs of ONE node, so the two gradient streams (through the row-normalised adjacency of the convolutions, and
    directly into A~ from ``A~ S`` of _diff_pool) meet inside the fused backward kernel instead of in an autograd add."""

    @staticmethod
    def forward(ctx, A, p):
        A = _f32c(A)
        b, c, _ = A.shape
        dev = A.device
        At = torch.empty_like(A) if p is not None else None
        An = torch.empty_like(A)
        invd = torch.empty(b * c, dtype=torch.float32, device=dev)
        ge1 = torch.empty(b * c, dtype=torch.float32, device=dev)
        K().adj_prep_fwd(A, b * c, c, p, At, An, invd, ge1)
        ctx.save_for_backward(A, An, invd, ge1)
        ctx.p = p
        if At is None:
            At = A.view_as(A)
        return At, An

    @staticmethod
    def backward(ctx, gAt, gAn):
        A, An, invd, ge1 = ctx.saved_tensors
        b, c, _ = A.shape
        if gAn is None:                                   # only the pass-through was used
            if ctx.p is None:
                return gAt, None
            gAn = torch.zeros_like(A)
        dA = torch.empty_like(A)
        K().adj_prep_bwd(A, An, invd, ge1, _f32c(gAn), None if gAt is None else _f32c(gAt), b * c, c, ctx.p, dA)
        return dA, None


def adj_prep(A, p=None):
    """Returns (A~, A_norm): A~ = _re_norm_adj(A, p) (model/network.py:183-191; A itself when p is None) and
    A_norm = A~ / clamp(rowsum(A~), min=1) (DenseSAGEConv's mean divisor folded into the adjacency)."""
    return _AdjPrep.apply(A, None if p is None else float(p))


def _bind_gemm_mode():
    """The GEMM mode of the per-operator path (kernels.GEMM_EXACT / GEMM_SPLIT_BF16) is a property of the ENCODER whose forward issued a
    node, but the kernel table reads it from a process-wide attribute at call time: every node records the mode its forward ran
    under and its backward re-installs that mode for its own duration -- two encoders in different modes (or a test toggling the
    mode between forward and backward) cannot end up with the forward and the backward of one model on different GEMM kernels.
    (The sequencer path carries the mode in cgc_level_desc.flags.)"""
    def bind(cls):
        fwd, bwd = cls.forward, cls.backward

        def forward(ctx, *args):
            ctx._gemm_mode = int(getattr(K(), 'gemm_mode', 0))
            return fwd(ctx, *args)

        def backward(ctx, *grads):
            k = K()
            old = getattr(k, 'gemm_mode', 0)
            k.gemm_mode = ctx._gemm_mode
            try:
                return bwd(ctx, *grads)
            finally:
                k.gemm_mode = old
        cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)

    for v in list(globals().values()):
        if isinstance(v, type) and issubclass(v, Function) and v is not Function:
            bind(v)


_bind_gemm_mode()
