"""Tensor-level front door of ``libcgc_hip.so`` (the C-ABI declared in include/cgc_hip.h).

``KernelSpec`` documents the exact arithmetic of every entry point -- it is the contract that the
HIP kernels (csrc/*.hip), the op-level checker (oracle/flat_ref.py, tests only) and the autograd
layer (ops.py) share.  ``HipKernels`` marshals torch tensors to raw device pointers and calls the
library through ctypes on torch's current HIP stream.  There is exactly one product implementation:
``get()`` raises if the library cannot be loaded or a tensor is not on the GPU.

All float tensors are fp32, all index tensors int32 unless stated; "ld" = leading dimension in
elements of a row-major matrix.  Kernels never allocate and never synchronise.
"""
import ctypes
import os

import torch

ACT_CODES = {'identity': 0, 'relu': 1, 'elu': 2, 'leakyrelu': 3}
L2_EPS = 1e-12       # F.normalize eps
RENORM_EPS = 1e-15   # model/network.py:8


class KernelSpec(object):
    """Semantics of the kernel entry points (all outputs are caller-allocated ``out`` tensors)."""

    # ------------------------------------------------------------------ graph structure (A1)
    def csr_build(self, edge_index, n, add_diag):
        """COO (int64 [2,E], row = aggregating centre) -> CSR + its transpose, both column-sorted.

        Duplicate edges collapse (model/utils.py:28-33 ASSIGNS 1); ``add_diag`` inserts (i,i) for
        every i (needed by edge_renorm, whose result always has a diagonal).  Returns a dict of
        int32 tensors: rowptr[n+1], col[cap], rowidx[cap], t_rowptr[n+1], t_col[cap], t_perm[cap]
        with cap = E (+n); entries at or beyond nnz = rowptr[n] are undefined.  t_col[k] is the
        source row of transposed slot k and t_perm[k] its slot in the forward arrays.  Edges with an id outside [0, n) are
        dropped and counted in ``bad_edges`` (int32 [1] on the device; the reference's dense indexing raises instead).
        """
        raise NotImplementedError

    def collate(self, x, mean, std, gptr, num_graphs, batch_out, edge_index, eptr):
        """Device-side finish of Batch.from_data_list: x [n,F] z-scored in place with mean/std [F] (None: untouched),
        batch_out int64 [n] = graph id per node (None: skipped), edge_index int64 [2,E] with per-graph local ids gets the
        node offset gptr[g] of its graph, the edges of graph g being [eptr[g], eptr[g+1]) (None: skipped)."""
        raise NotImplementedError

    def farthest_point_sample(self, pos, gptr, num_graphs, max_nodes, start, optr, out, table16=False):
        """Farthest-point sampling per graph (FarthestSampler, common/utils.py:187-197, on coordinates instead of the
        distance table): pos [n,2] f32, gptr int32 [B+1], start int32 [B] first pick (local index), optr int32 [B+1]
        offsets of the picks, out int32 [optr[B]] global node ids in pick order; ties -> lowest index.
        table16=False: squared fp64 distances.  table16=True: the reference's table entries int16(sqrt(dx^2+dy^2)) in
        float32 (dataflow/construct_feature_graph.py:17-24) -- the reference's picks index for index."""
        raise NotImplementedError

    def radius_knn(self, pos, gptr, num_graphs, r, k, loop):
        """Cell-graph construction for a batch of graphs (torch_cluster.radius_graph(pos, r, None, loop, k) per graph,
        dataflow/data.py:348): pos [n,2] f32, gptr int32 [B+1].  Per node its <= k nearest others within r (+ itself iff
        loop), restricted to its own graph.  Returns edge_index int64 [2, nnz] (row = centre, ascending; neighbours by
        distance, ties by index) with global node ids."""
        raise NotImplementedError

    def edge_renorm(self, rowptr, col, n, p, val_out):
        """Level-1 ``_re_norm_adj`` (model/network.py:183-191) on a 0/1 CSR that holds its diagonal:
        val[k] = p if col[k]==row else (1/(c+1e-15))*(1-p), c = number of off-diagonal entries of the row."""
        raise NotImplementedError

    def csr_transpose_vals(self, t_rowptr, t_perm, val, n, t_val_out):
        """t_val[k] = val[t_perm[k]] for the live slots of the transposed CSR (entries beyond nnz stay undefined)."""
        raise NotImplementedError

    def csr_invdeg(self, rowptr, val, n, out):
        """out[i] = 1 / max(sum_k val[k] (or the entry count when val is None), 1)  -- DenseSAGEConv's clamp."""
        raise NotImplementedError

    def spmm(self, rowptr, col, perm, val, pre, post, x, out, n, width, gptr=None, num_graphs=0, nmax=0, visit=0, ld=None, gorder=None):
        """out[i,:] = post[i] * sum_{k in row i} w_k * pre[col[k]] * x[col[k],:]
        with w_k = val[perm[k]] / val[k] / 1 and pre/post optional.  x, out: [n, width] contiguous.
        gptr/num_graphs/nmax (optional): the rows are a batch of graphs with block-diagonal adjacency
        (first row of each graph, count, largest graph) -- a layout hint, the result is the same.
        visit (optional, scheduling hint): 1 = x was just written in ascending row order, 2 = by a ragged batched gemm.
        ld (optional): row stride of x and out when the rows are padded (wide rows only; needs gptr).
        gorder (optional, scheduling only): int32 [num_graphs] visiting sequence of the graphs (wide rows)."""
        raise NotImplementedError

    # ------------------------------------------------------------------ dense contractions (MFMA fp32)
    def gemm(self, A, B, C, M, N, K, transA, transB, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None,
             batch=1, strideA=0, strideB=0, strideC=0, gptr=None, ragged=0, max_ragged=0, ragged_total=0, extra=()):
        """C_b = alpha * op(A_b) op(B_b) + beta * C_b (+ bias[N]),  b = 0..batch-1, row-major.

        op(A) is M x K (stored [M,K] or, transA, [K,M]); op(B) is K x N (stored [K,N] or, transB, [N,K]).
        Operand b starts at base + b*stride (+ ragged offset).  ragged=1: M_b = gptr[b+1]-gptr[b], A
        (not transposed) and C advance by gptr[b] rows.  ragged=2: K_b = gptr[b+1]-gptr[b], A (transposed)
        and B (not transposed) advance by gptr[b] rows.  max_ragged bounds the ragged extent (grid size);
        ragged_total = sum of the ragged extents (host-side flop accounting only).
        extra: up to two (A_x, B_x, lda, ldb, K_x, strideA, strideB) pairs whose products are added to op(A)op(B)
        (same orientation, M, N, batch and ragged row offsets): the concatenated-K product without the concatenation.
        """
        raise NotImplementedError

    def reduce_batch_sum(self, ws, out, parts, numel, beta=0.0):
        """out[j] = beta*out[j] + sum_s ws[s*numel + j]  (deterministic split-K combine)."""
        raise NotImplementedError

    def reduce_batched(self, ws, out, outer, parts, numel, beta=0.0):
        """out[o, j] = beta*out[o, j] + sum_s ws[o, s, j]  for ws [outer, parts, numel]."""
        raise NotImplementedError

    # ------------------------------------------------------------------ conv epilogue: L2 norm, activation, BatchNorm (A4, A5)
    def l2norm_act_stats(self, h, n, F, normalize, act, hn_out, rinv_out, stats_out):
        """hn = h / max(||h||_2, 1e-12) row-wise (or hn = h), rinv = that reciprocal;
        stats[0,f] = sum_i act(hn)[i,f], stats[1,f] = sum_i act(hn)[i,f]^2 (None to skip).  ``stats`` is
        FLOAT64 [2,F]: the variance is a difference of these two sums."""
        raise NotImplementedError

    def l2norm_act_bn(self, h, n, F, normalize, act, hn_out, rinv_out, count, eps, momentum, running_mean, running_var,
                      num_batches_tracked, mean_out, istd_out):
        """l2norm_act_stats + bn_finalize as one call (the training forward): also increments num_batches_tracked
        (int64 scalar tensor or None), as nn.BatchNorm1d.forward does."""
        raise NotImplementedError

    def sage_wide_fwd(self, agg, lda, weight, bias, n, Kin, F, normalize, act, hn_out, rinv_out, stats, count, eps, momentum,
                      running_mean, running_var, num_batches_tracked, mean_out, istd_out):
        """hn = l2norm(agg[:, :Kin] @ weight + bias) and (stats) the statistics part of l2norm_act_bn, as one kernel: narrow
        inputs (Kin <= 32) into narrow (F <= 32) or wide (F <= 1664) outputs.  Returns False (nothing done) outside that envelope."""
        raise NotImplementedError

    def bn_finalize(self, stats, count, eps, momentum, running_mean, running_var, mean_out, istd_out):
        """mean = s0/count, var = s1/count - mean^2 (biased), istd = rsqrt(var+eps); running stats get
        momentum updates with the unbiased var*count/(count-1).  ``count`` = B*Nmax INCLUDING the
        zero padding rows of the dense layout (model/network.py:101-107)."""
        raise NotImplementedError

    def bn_act_apply(self, hn, n, F, act, mean, istd, gamma, beta, y_out, ldy):
        """y = (act(hn) - mean) * istd * gamma + beta   (mean None -> y = act(hn))."""
        raise NotImplementedError

    def bn_bwd_reduce(self, dy, ldy, hn, n, F, act, mean, istd, sums_out):
        """sums[0,f] = sum_i dy, sums[1,f] = sum_i dy * xhat, xhat = (act(hn)-mean)*istd."""
        raise NotImplementedError

    def bn_act_l2_bwd(self, dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, dh_out,
                      dh_colsum_out=None):
        """Backward of BN o act o l2norm for one row block (dh_colsum_out[f] = sum_i dh[i,f], optional).
        mode 2 (batch stats): do = gamma*istd*(dy - s0/count - xhat*s1/count); mode 1 (running stats):
        do = gamma*istd*dy; mode 0 (no BN): do = dy.   dhn = do*act'(hn);
        normalize: dh = rinv*(dhn - hn*<hn,dhn>) (dh = dhn*1e12 where the norm clamp was active)."""
        raise NotImplementedError

    def colsum(self, x, ld, n, F, out):
        """out[f] = sum_i x[i,f]."""
        raise NotImplementedError

    def sage_narrow_bwd(self, dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, agg, lda, fin, weight,
                        dagg_out, dwdb_out):
        """bn_act_l2_bwd without writing dh, plus its three consumers: dagg_out [n,fin] = dh @ weight^T (None: skipped) and
        dwdb_out [fin*F + F] = (agg^T dh).ravel() followed by colsum(dh).  Returns False (nothing done) unless fin, F <= 32."""
        raise NotImplementedError

    # ------------------------------------------------------------------ assignment softmax (A8), readout (A9)
    def softmax_fwd(self, x, n, C, out, ld=None):
        raise NotImplementedError

    def softmax_bwd(self, S, dS, n, C, dx_out, dx_colsum_out=None, ld=None):
        """dx = S * (dS - <dS, S>_row);  dx_colsum_out[c] = sum_i dx[i,c] (optional)."""
        raise NotImplementedError

    def segment_max_fwd(self, x, gptr, B, D, nmax, out, arg_out):
        """Per graph b and column d: max over its rows (first index on ties); a graph with fewer than
        nmax rows also competes against the zero padding rows of the dense layout (model/network.py:264):
        if its max is < 0 the result is 0 and arg = -1."""
        raise NotImplementedError

    def segment_max_bwd(self, dout, arg, B, D, dx_zeroed):
        """dx[arg[b,d], d] = dout[b,d] where arg >= 0 (dx arrives zero-filled)."""
        raise NotImplementedError

    def segment_max_bwd_full(self, dout, arg, gptr, B, D, nmax, dx_out):
        """segment_max_bwd that writes every element of dx (no pre-zeroed buffer needed)."""
        raise NotImplementedError

    # ------------------------------------------------------------------ jumping-knowledge attention (A7)
    def jk_supported(self, C):
        """Whether the fused DenseJK kernels exist for this channel count."""
        raise NotImplementedError

    def jk_fwd(self, xs, n, npad, C, lstm, w_att, b_att, out, HS, CS):
        """DenseJK forward (model/network.py:36-52): xs [n,3C] -> out [n,C].  ``lstm`` = 8 tensors
        (w_ih, w_hh, b_ih, b_hh) x (forward, reverse) in torch.nn.LSTM layout, hidden H = 3C/2.
        HS, CS [6H, npad]: hidden / cell state of (direction d, step t, unit j) at row (d*3+t)*H + j."""
        raise NotImplementedError

    def jk_bwd(self, xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, DGT, INT, DHC):
        """DenseJK backward: dxs [n,3C]; DGT [2, 4H+1, 3*npad] and INT [2, C+2H+1, 3*npad] such that
        G_d = DGT[d] @ INT[d]^T gives dW_ih = G_d[:4H,:C], dW_hh = G_d[:4H,C:C+H], db_ih = db_hh = G_d[:4H,C+H],
        d w_att[dH:(d+1)H] = G_d[4H, C+H+1:], d b_att = G_0[4H, C+H].  DHC [2,2,H,npad] is scratch."""
        raise NotImplementedError

    def jk_bwd_params(self, xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, G_out):
        """DenseJK backward with the parameter gradients delivered directly: dxs [n,3C] and G_out [2, 4H+1, C+2H+1] with
        the meaning of ``DGT[d] @ INT[d]^T`` in ``jk_bwd`` (entries of the last row that are not parameter gradients are
        unspecified)."""
        raise NotImplementedError

    def jk_unpack_param_grads(self, G, C):
        """G [2, 4H+1, C+2H+1] -> list of CONTIGUOUS gradients in parameter order: (w_ih, w_hh, b_ih, b_hh) forward, the same
        four reverse, att.weight [1, 2H], att.bias [1] -- all slices of one flat buffer."""
        raise NotImplementedError

    # ------------------------------------------------------------------ dense adjacency ops at levels 2-3 (A4, A6)
    def dense_rownorm_fwd(self, A, R, C, out, invd_out, ge1_out):
        """s = rowsum(A); d = max(s,1); out = A/d; invd = 1/d; ge1 = (s >= 1)  (clamp(min=1) of DenseSAGEConv)."""
        raise NotImplementedError

    def dense_rownorm_bwd(self, dOut, Anorm, invd, ge1, R, C, dA_out):
        """dA = invd * (dOut - ge1 * <dOut, Anorm>_row)."""
        raise NotImplementedError

    def dense_renorm_fwd(self, A, R, C, p, out):
        """``_re_norm_adj`` on [B,C,C] viewed as [R=B*C, C]; the diagonal of row i is column i mod C."""
        raise NotImplementedError

    def dense_renorm_bwd(self, A, dOut, R, C, p, dA_out):
        raise NotImplementedError

    def adj_prep_fwd(self, A, R, C, p, At_out, An_out, invd_out, ge1_out):
        """dense_renorm_fwd (skipped when p is None: At_out None) followed by dense_rownorm_fwd of its result, one pass."""
        raise NotImplementedError

    def adj_prep_bwd(self, A, An, invd, ge1, gAn, gAt, R, C, p, dA_out):
        """dAt = dense_rownorm_bwd(gAn) + gAt (gAt may be None); dA = dense_renorm_bwd(A, dAt) (dA = dAt when p is None)."""
        raise NotImplementedError


# ----------------------------------------------------------------------------------------------
_LIB_NAME = 'libcgc_hip.so'
_instance = None


def lib_path():
    # CGC_LIB: another build of the same library (A/B timing of kernel changes on one box); default: the in-tree build
    return os.environ.get('CGC_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', _LIB_NAME)


def get():
    """The process-wide kernel table.  Raises (never falls back) when the HIP library is unusable."""
    global _instance
    if _instance is None:
        _instance = HipKernels()
    return _instance


_EINVAL = -1                            # include/cgc_hip.h: CGC_EINVAL (nothing was launched)
GEMM_EXACT, GEMM_SPLIT_BF16, GEMM_SPLIT_F16 = 0, 1, 2      # include/cgc_hip.h: CGC_GEMM_EXACT / CGC_GEMM_SPLIT_BF16 / CGC_GEMM_SPLIT_F16


def is_native():
    return isinstance(_instance, HipKernels)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)
if _raw_stream is None or _cur_device is None:          # older / newer torch without the private accessors
    def _raw_stream(_idx):
        return torch.cuda.current_stream().cuda_stream

    def _cur_device():
        return torch.cuda.current_device()


def split_jk_param_grads(flat, C):
    """Views of the flat DenseJK parameter-gradient buffer (layout: cgc_jk_unpack_param_grads, include/cgc_hip.h)."""
    H = 3 * C // 2
    out, o = [], 0
    for _ in range(2):
        for shape in ((4 * H, C), (4 * H, H), (4 * H,), (4 * H,)):
            k = 1
            for d in shape:
                k *= d
            out.append(flat[o:o + k].view(shape))
            o += k
    out.append(flat[o:o + 2 * H].view(1, 2 * H))
    out.append(flat[o + 2 * H:o + 2 * H + 1])
    return out


def _ptr(t):
    # a plain int (None = NULL): every prototype is declared (_abi.py), so ctypes converts it -- building a c_void_p object per
    # argument was ~0.1 ms of a 4 ms step (~950 pointer arguments per step)
    return t.data_ptr() if t is not None else None


class LaunchTimer(object):
    """HIP-event timing of every launch of the dominant 128 x 128 GEMM and of the wide SpMM between ``start()`` and ``stop()``
    (the library's measurement hook, csrc/timing.hip: events on the stream of the launch, whoever issues it -- the per-operator
    path or the step sequencer).  ``records()`` after a synchronize: list of (tag, dims, ms)."""
    TAGS = {1: 'gemm_128x128', 2: 'spmm_wide'}

    def __init__(self, capacity=8192):
        self.lib = get().lib
        self.h = self.lib.cgc_timing_create(int(capacity))
        if not self.h:
            raise RuntimeError('cgc_timing_create failed')

    def start(self):
        self.lib.cgc_timing_attach(self.h)
        return self

    def stop(self):
        self.lib.cgc_timing_attach(None)

    def records(self):
        out = []
        dims, ms = (ctypes.c_int * 8)(), ctypes.c_float()
        for i in range(self.lib.cgc_timing_count(self.h)):
            rc = self.lib.cgc_timing_read(self.h, i, dims, ctypes.byref(ms))
            if rc != 0:
                raise RuntimeError('cgc_timing_read failed with code %d' % rc)
            out.append((self.TAGS.get(dims[0], str(dims[0])), tuple(dims[1:8]), float(ms.value)))
        return out

    def counts(self):
        c = {}
        for i in range(self.lib.cgc_timing_count(self.h)):
            dims, ms = (ctypes.c_int * 8)(), ctypes.c_float()
            self.lib.cgc_timing_read(self.h, i, dims, ctypes.byref(ms))
            c[self.TAGS.get(dims[0], str(dims[0]))] = c.get(self.TAGS.get(dims[0], str(dims[0])), 0) + 1
        return c

    def close(self):
        if self.h:
            self.lib.cgc_timing_destroy(self.h)
            self.h = None

    @staticmethod
    def gemm_flops(dims, ragged_total):
        """Algorithmic flops 2 M N K of a recorded GEMM launch; ``ragged_total`` = the rows the ragged extents add up to (the
        batch's node count: the library does not know it)."""
        M, N, K, batch, ragged, _, xk = dims
        if ragged == 1:
            return 2.0 * ragged_total * N * (K + xk)
        if ragged == 2:
            return 2.0 * M * N * ragged_total
        if ragged == 3:
            parts = -(-K // dims[5])
            return 2.0 * M * N * K * (batch // parts)
        return 2.0 * M * N * (K + xk) * batch


class _NoWorkspace(object):
    @staticmethod
    def data_ptr():
        return None

    @staticmethod
    def numel():
        return 0


_NO_WS = _NoWorkspace()


class HipKernels(KernelSpec):
    tail_split = True   # hand cgc_gemm_f32_ws its slab workspace (False: every output tile is computed whole; tests / A-B timing)
    # cgc_gemm_f32_ws's `mode` for the products issued through this table (the per-operator path): GEMM_EXACT (default),
    # GEMM_SPLIT_BF16 -- the big products as six bf16 MFMA pairs per fp32 product (csrc/gemm_split.hip) -- or GEMM_SPLIT_F16 -- as
    # three fp16 pairs of operands scaled per batch item (csrc/gemm_half.hip).  The encoder sets it from its own ``gemm_mode`` at
    # the top of forward(); the sequencer gets the same choice through cgc_level_desc.flags bits 1 / 2.
    gemm_mode = 0
    # graph structure graph by graph in two launches when the Batch says how its edge list is grouped (cgc_graph_build_local;
    # CGC_GRAPH_LOCAL=0 / False: always the general build -- A-B timing, tests)
    graph_local = os.environ.get('CGC_GRAPH_LOCAL', '1') != '0'
    graph_local_count = 0      # batches built that way by this process (tests: did the route apply?)

    def __init__(self):
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError('%s not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(there is no CPU fallback)' % path)
        if not torch.cuda.is_available():
            raise RuntimeError('cgc_net_amd needs an AMD GPU (gfx950); torch.cuda.is_available() is False '
                               'and there is no CPU fallback')
        self.lib = ctypes.CDLL(path)
        from . import _abi
        _abi.declare(self.lib)
        self._ws_cache = {}
        self._graph_local_max = int(self.lib.cgc_graph_local_max_nodes())

    # -- helpers
    @staticmethod
    def _stream():
        # torch's current stream of the current device as a raw hipStream_t (the C call: torch.cuda.current_stream() costs
        # ~8 us of Python per launch, which is what bounds the small-graph regime)
        return _raw_stream(_cur_device())

    @staticmethod
    def _chk(rc, name):
        if rc != 0:
            raise RuntimeError('%s failed with code %d' % (name, rc))

    @staticmethod
    def _dev(*ts):
        # kernels are enqueued on the CURRENT device's current stream (_stream()): tensors living on another GPU would be
        # touched from the wrong device's stream, unordered against torch's work on them -- refuse instead of racing
        cur = _cur_device()
        for t in ts:
            if t is not None:
                if not t.is_cuda:
                    raise RuntimeError('cgc_net_amd kernels take GPU tensors only (got a %s tensor)' % t.device)
                if t.get_device() != cur:
                    raise RuntimeError('tensor on cuda:%d but the current device is cuda:%d: wrap the call in '
                                       'torch.cuda.device(...) / call torch.cuda.set_device first' % (t.get_device(), cur))

    # -- graph structure
    def csr_build(self, edge_index, n, add_diag):
        self._dev(edge_index)
        edge_index = edge_index.to(torch.int64).contiguous()
        E = edge_index.shape[1]
        cap = E + (n if add_diag else 0)
        dev = edge_index.device
        i32 = dict(dtype=torch.int32, device=dev)
        out = {k: torch.empty(n + 1, **i32) for k in ('rowptr', 't_rowptr')}
        out.update({k: torch.empty(max(cap, 1), **i32) for k in ('col', 'rowidx', 't_col', 't_perm')})
        ws = torch.empty(3 * (n + 1) + 2 * max(cap, 1), **i32)
        rc = self.lib.cgc_csr_build(_ptr(edge_index), ctypes.c_int64(E), n, int(add_diag),
                                    _ptr(out['rowptr']), _ptr(out['col']), _ptr(out['rowidx']),
                                    _ptr(out['t_rowptr']), _ptr(out['t_col']), _ptr(out['t_perm']),
                                    _ptr(ws), self._stream())
        self._chk(rc, 'cgc_csr_build')
        out['cap'] = cap
        o = int(self.lib.cgc_csr_bad_edges_offset(ctypes.c_int64(E), n, int(add_diag)))
        out['bad_edges'] = ws[o:o + 1]              # device-side count of dropped out-of-range edges (no sync here)
        return out

    def graph_build(self, edge_index, n, renorm_p, gptr=None, eptr=None, num_graphs=0, nmax=0, emax=0):
        """csr_build (+ edge_renorm + csr_transpose_vals when renorm_p is not None) + csr_invdeg behind ONE library call, all
        outputs carved out of two allocations.  Returns the dict of csr_build plus val / t_val (None without renorm) and inv_d.
        With ``gptr`` / ``eptr`` (int32 [B+1] on the device: node and edge ranges of the graphs, the edge list grouped by graph as
        Batch.from_data_list emits it; ``nmax`` / ``emax``: the largest graph's nodes, the most edges of one graph) the structure is built
        graph by graph in two launches (cgc_graph_build_local: same arrays bit for bit); batches outside its envelope take the general
        build."""
        self._dev(edge_index)
        edge_index = edge_index.to(torch.int64).contiguous()
        E = edge_index.shape[1]
        renorm = renorm_p is not None
        cap = max(E + (n if renorm else 0), 1)
        dev = edge_index.device
        a = lambda k: -(-k // 64) * 64                    # 256-byte aligned pieces
        sizes = [n + 1, n + 1, cap, cap, cap, cap, 3 * (n + 1) + 2 * cap]
        ibuf = torch.empty(sum(a(k) for k in sizes), dtype=torch.int32, device=dev)
        parts, o = [], 0
        for k in sizes:
            parts.append(ibuf[o:o + k])
            o += a(k)
        rowptr, t_rowptr, col, rowidx, t_col, t_perm, ws = parts
        fbuf = torch.empty((2 * a(cap) if renorm else 0) + max(n, 1), dtype=torch.float32, device=dev)
        val = fbuf[:cap] if renorm else None
        t_val = fbuf[a(cap):a(cap) + cap] if renorm else None
        inv_d = fbuf[2 * a(cap):] if renorm else fbuf
        rc = _EINVAL
        if gptr is not None and eptr is not None and self.graph_local and 0 < nmax <= self._graph_local_max:
            self._dev(gptr, eptr)
            rc = self.lib.cgc_graph_build_local(_ptr(edge_index), ctypes.c_int64(E), n, _ptr(gptr), _ptr(eptr), int(num_graphs), int(nmax), int(emax),
                                                ctypes.c_float(-1.0 if renorm_p is None else renorm_p), _ptr(rowptr), _ptr(col), _ptr(rowidx),
                                                _ptr(t_rowptr), _ptr(t_col), _ptr(t_perm), _ptr(val), _ptr(t_val), _ptr(inv_d), _ptr(ws),
                                                self._stream())
            if rc != _EINVAL:
                self._chk(rc, 'cgc_graph_build_local')
                self.graph_local_count += 1
        if rc == _EINVAL:                                  # no graph ranges, or outside the graph-local build's envelope
            self._chk(self.lib.cgc_graph_build(_ptr(edge_index), ctypes.c_int64(E), n, ctypes.c_float(-1.0 if renorm_p is None else renorm_p),
                                               _ptr(rowptr), _ptr(col), _ptr(rowidx), _ptr(t_rowptr), _ptr(t_col), _ptr(t_perm), _ptr(val),
                                               _ptr(t_val), _ptr(inv_d), _ptr(ws), self._stream()), 'cgc_graph_build')
        bo = int(self.lib.cgc_csr_bad_edges_offset(ctypes.c_int64(E), n, int(renorm)))
        return dict(rowptr=rowptr, t_rowptr=t_rowptr, col=col, rowidx=rowidx, t_col=t_col, t_perm=t_perm, cap=E + (n if renorm else 0),
                    bad_edges=ws[bo:bo + 1], val=val, t_val=t_val, inv_d=inv_d)

    def collate(self, x, mean, std, gptr, num_graphs, batch_out, edge_index, eptr):
        self._dev(x, mean, std, gptr, batch_out, edge_index, eptr)
        assert x.is_contiguous() and x.dtype == torch.float32
        E = edge_index.shape[1] if edge_index is not None else 0
        assert edge_index is None or (edge_index.is_contiguous() and edge_index.dtype == torch.int64)
        self._chk(self.lib.cgc_collate(_ptr(x), x.shape[0], x.shape[1], _ptr(mean), _ptr(std), _ptr(gptr), num_graphs,
                                       _ptr(batch_out), _ptr(edge_index), ctypes.c_int64(E), _ptr(eptr), self._stream()),
                  'cgc_collate')

    def farthest_point_sample(self, pos, gptr, num_graphs, max_nodes, start, optr, out, table16=False):
        self._dev(pos, gptr, start, optr, out)
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[1] == 2
        fn = self.lib.cgc_farthest_point_sample_table16 if table16 else self.lib.cgc_farthest_point_sample
        self._chk(fn(_ptr(pos), _ptr(gptr), num_graphs, max_nodes, _ptr(start), _ptr(optr), _ptr(out), self._stream()),
                  'cgc_farthest_point_sample')

    def radius_knn(self, pos, gptr, num_graphs, r, k, loop):
        self._dev(pos, gptr)
        pos = pos.to(torch.float32).contiguous()
        n, dev = pos.shape[0], pos.device
        i32 = dict(dtype=torch.int32, device=dev)
        if n == 0:
            return torch.zeros(2, 0, dtype=torch.int64, device=dev)
        assert pos.dim() == 2 and pos.shape[1] == 2 and gptr.dtype == torch.int32
        nbr, cnt, rowptr = torch.empty(n, k + 1, **i32), torch.empty(n, **i32), torch.empty(n + 1, **i32)
        ws = torch.empty(int(self.lib.cgc_radius_knn_ws_ints(n, num_graphs)), **i32)
        self._chk(self.lib.cgc_radius_knn(_ptr(pos), _ptr(gptr), num_graphs, n, ctypes.c_float(r), k, int(bool(loop)),
                                          _ptr(nbr), _ptr(cnt), _ptr(rowptr), _ptr(ws), self._stream()), 'cgc_radius_knn')
        nnz = int(rowptr[n].item())                    # the one host sync of graph construction: the edge count
        ei = torch.empty(2, nnz, dtype=torch.int64, device=dev)
        self._chk(self.lib.cgc_knn_emit_edges(_ptr(nbr), _ptr(rowptr), n, k, ctypes.c_int64(nnz), _ptr(ei), self._stream()),
                  'cgc_knn_emit_edges')
        return ei

    def edge_renorm(self, rowptr, col, n, p, val_out):
        self._dev(rowptr, col, val_out)
        self._chk(self.lib.cgc_edge_renorm(_ptr(rowptr), _ptr(col), n, ctypes.c_float(p), _ptr(val_out),
                                           self._stream()), 'cgc_edge_renorm')

    def csr_transpose_vals(self, t_rowptr, t_perm, val, n, t_val_out):
        self._dev(t_rowptr, t_perm, val, t_val_out)
        self._chk(self.lib.cgc_csr_transpose_vals(_ptr(t_rowptr), _ptr(t_perm), _ptr(val), n, _ptr(t_val_out), self._stream()),
                  'cgc_csr_transpose_vals')

    def csr_invdeg(self, rowptr, val, n, out):
        self._dev(rowptr, val, out)
        self._chk(self.lib.cgc_csr_invdeg(_ptr(rowptr), _ptr(val), n, _ptr(out), self._stream()), 'cgc_csr_invdeg')

    def spmm(self, rowptr, col, perm, val, pre, post, x, out, n, width, gptr=None, num_graphs=0, nmax=0, visit=0, ld=None, gorder=None):
        self._dev(rowptr, col, perm, val, pre, post, x, out, gptr)
        assert (x.is_contiguous() and out.is_contiguous()) if ld is None else (gptr is not None and x.stride(1) == 1 and out.stride(1) == 1)
        if gptr is not None:
            self._chk(self.lib.cgc_spmm_graphs_ordered(_ptr(rowptr), _ptr(col), _ptr(perm), _ptr(val), _ptr(pre), _ptr(post),
                                                       _ptr(x), _ptr(out), n, width, width if ld is None else ld, _ptr(gptr),
                                                       num_graphs, nmax, int(visit), _ptr(gorder), self._stream()), 'cgc_spmm_graphs')
        else:
            self._chk(self.lib.cgc_spmm(_ptr(rowptr), _ptr(col), _ptr(perm), _ptr(val), _ptr(pre), _ptr(post),
                                        _ptr(x), _ptr(out), n, width, self._stream()), 'cgc_spmm')

    # -- dense contractions
    def _gemm_ws(self, device, stream):
        """The workspace of cgc_gemm_f32_ws (include/cgc_hip.h): the slabs of the tail split and, at its end, the scale slots of mode
        GEMM_SPLIT_F16; one per (device, stream): products queued on one stream run one after the other and may share it; two
        streams must not.  tail_split = False: no slabs -- mode GEMM_SPLIT_F16 still gets its scale slots (a workspace too small for
        a slab)."""
        if not self.tail_split and int(self.gemm_mode) != GEMM_SPLIT_F16:
            return _NO_WS
        key = (device.index, stream)
        ws = self._ws_cache.get(key)
        if ws is None:
            ws = self._ws_cache[key] = torch.empty(int(self.lib.cgc_gemm_ws_floats()), dtype=torch.float32, device=device)
        return ws if self.tail_split else ws[-int(self.lib.cgc_gemm_half_ws_floats()):]

    def gemm(self, A, B, C, M, N, K, transA, transB, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None,
             batch=1, strideA=0, strideB=0, strideC=0, gptr=None, ragged=0, max_ragged=0, ragged_total=0, extra=()):
        self._dev(A, B, C, bias, gptr)
        stream = self._stream()
        ws = self._gemm_ws(C.device, stream)
        if extra:
            nx = len(extra)
            self._dev(*[e[0] for e in extra], *[e[1] for e in extra])
            PA, IA, LA = ctypes.c_void_p * nx, ctypes.c_int * nx, ctypes.c_int64 * nx
            rc = self.lib.cgc_gemm_f32_cat_ws(int(transA), int(transB), M, N, K, ctypes.c_float(alpha), _ptr(A), lda,
                                              _ptr(B), ldb, ctypes.c_float(beta), _ptr(C), ldc, _ptr(bias), batch,
                                              ctypes.c_int64(strideA), ctypes.c_int64(strideB), ctypes.c_int64(strideC),
                                              _ptr(gptr), ragged, max_ragged, nx,
                                              PA(*[e[0].data_ptr() for e in extra]), IA(*[e[2] for e in extra]),
                                              LA(*[e[5] for e in extra]), PA(*[e[1].data_ptr() for e in extra]),
                                              IA(*[e[3] for e in extra]), LA(*[e[6] for e in extra]),
                                              IA(*[e[4] for e in extra]), ws.data_ptr(), ws.numel(), int(self.gemm_mode), stream)
        else:
            rc = self.lib.cgc_gemm_f32_ws(int(transA), int(transB), M, N, K, ctypes.c_float(alpha), _ptr(A), lda,
                                          _ptr(B), ldb, ctypes.c_float(beta), _ptr(C), ldc, _ptr(bias), batch,
                                          ctypes.c_int64(strideA), ctypes.c_int64(strideB), ctypes.c_int64(strideC),
                                          _ptr(gptr), ragged, max_ragged, ws.data_ptr(), ws.numel(), int(self.gemm_mode), stream)
        self._chk(rc, 'cgc_gemm_f32')

    def reduce_batch_sum(self, ws, out, parts, numel, beta=0.0):
        self._dev(ws, out)
        self._chk(self.lib.cgc_reduce_batch_sum(_ptr(ws), _ptr(out), parts, ctypes.c_int64(numel),
                                                ctypes.c_float(beta), self._stream()), 'cgc_reduce_batch_sum')

    def reduce_batched(self, ws, out, outer, parts, numel, beta=0.0):
        self._dev(ws, out)
        self._chk(self.lib.cgc_reduce_batched(_ptr(ws), _ptr(out), outer, parts, numel, ctypes.c_float(beta),
                                              self._stream()), 'cgc_reduce_batched')

    # -- conv epilogue
    def l2norm_act_stats(self, h, n, F, normalize, act, hn_out, rinv_out, stats_out):
        self._dev(h, hn_out, rinv_out, stats_out)
        ws = torch.empty(int(self.lib.cgc_stats_ws_floats(n, F)), dtype=torch.float32, device=h.device) if stats_out is not None else None
        self._chk(self.lib.cgc_l2norm_act_stats(_ptr(h), n, F, int(normalize), act, _ptr(hn_out), _ptr(rinv_out),
                                                _ptr(stats_out), _ptr(ws), self._stream()), 'cgc_l2norm_act_stats')

    def sage_wide_fwd(self, agg, lda, weight, bias, n, Kin, F, normalize, act, hn_out, rinv_out, stats, count, eps, momentum,
                      running_mean, running_var, num_batches_tracked, mean_out, istd_out):
        self._dev(agg, weight, bias, hn_out, rinv_out, running_mean, running_var, num_batches_tracked, mean_out, istd_out)
        ws = None
        if stats:
            ws = torch.empty(int(self.lib.cgc_stats_ws_floats(n, F)), dtype=torch.float32, device=agg.device)
        tail = (int(bool(stats)), _ptr(ws), ctypes.c_double(count), ctypes.c_float(eps), ctypes.c_float(momentum), _ptr(running_mean),
                _ptr(running_var), _ptr(num_batches_tracked), _ptr(mean_out), _ptr(istd_out), self._stream())
        if F <= 32 and hn_out.stride(0) == F:          # narrow output: one wave per 32 rows (csrc/sagenarrow.hip)
            rc = self.lib.cgc_sage_narrow_fwd(_ptr(agg), lda, _ptr(weight), _ptr(bias), n, Kin, F, int(normalize), act, _ptr(hn_out),
                                              _ptr(rinv_out), *tail)
        else:
            rc = self.lib.cgc_sage_wide_fwd(_ptr(agg), lda, _ptr(weight), _ptr(bias), n, Kin, F, int(normalize), act, _ptr(hn_out),
                                            hn_out.stride(0), _ptr(rinv_out), *tail)
        if rc == -1:
            return False
        self._chk(rc, 'cgc_sage_wide_fwd')
        return True

    def bn_finalize(self, stats, count, eps, momentum, running_mean, running_var, mean_out, istd_out):
        self._dev(stats, running_mean, running_var, mean_out, istd_out)
        F = mean_out.numel()
        self._chk(self.lib.cgc_bn_finalize(_ptr(stats), F, ctypes.c_double(count), ctypes.c_float(eps),
                                           ctypes.c_float(momentum), _ptr(running_mean), _ptr(running_var),
                                           _ptr(mean_out), _ptr(istd_out), self._stream()), 'cgc_bn_finalize')

    def bn_act_apply(self, hn, n, F, act, mean, istd, gamma, beta, y_out, ldy):
        self._dev(hn, mean, istd, gamma, beta, y_out)
        self._chk(self.lib.cgc_bn_act_apply(_ptr(hn), n, F, act, _ptr(mean), _ptr(istd), _ptr(gamma), _ptr(beta),
                                            _ptr(y_out), ldy, self._stream()), 'cgc_bn_act_apply')

    def l2norm_act_bn(self, h, n, F, normalize, act, hn_out, rinv_out, count, eps, momentum, running_mean, running_var,
                      num_batches_tracked, mean_out, istd_out):
        self._dev(h, hn_out, rinv_out, running_mean, running_var, num_batches_tracked, mean_out, istd_out)
        ws = torch.empty(int(self.lib.cgc_stats_ws_floats(n, F)), dtype=torch.float32, device=h.device)
        assert num_batches_tracked is None or num_batches_tracked.dtype == torch.int64
        self._chk(self.lib.cgc_l2norm_act_bn(_ptr(h), n, F, int(normalize), act, _ptr(hn_out), _ptr(rinv_out), _ptr(ws),
                                             ctypes.c_double(count), ctypes.c_float(eps),
                                             ctypes.c_float(momentum), _ptr(running_mean), _ptr(running_var),
                                             _ptr(num_batches_tracked), _ptr(mean_out), _ptr(istd_out), self._stream()),
                  'cgc_l2norm_act_bn')

    def bn_bwd_reduce(self, dy, ldy, hn, n, F, act, mean, istd, sums_out):
        self._dev(dy, hn, mean, istd, sums_out)
        nblk = self.lib.cgc_stats_blocks(n, F)
        ws = torch.empty(max(nblk, 1) * 2 * F, dtype=torch.float32, device=hn.device)
        self._chk(self.lib.cgc_bn_bwd_reduce(_ptr(dy), ldy, _ptr(hn), n, F, act, _ptr(mean), _ptr(istd),
                                             _ptr(sums_out), _ptr(ws), self._stream()), 'cgc_bn_bwd_reduce')

    def _slots(self, n, F, dev):
        return torch.empty(max(self.lib.cgc_stats_blocks(n, F), 1) * 2 * F, dtype=torch.float32, device=dev)

    def bn_act_l2_bwd(self, dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, dh_out,
                      dh_colsum_out=None):
        self._dev(dy, hn, rinv, mean, istd, gamma, sums, dh_out, dh_colsum_out)
        ws = self._slots(n, F, hn.device) if dh_colsum_out is not None else None
        self._chk(self.lib.cgc_bn_act_l2_bwd(_ptr(dy), ldy, _ptr(hn), _ptr(rinv), n, F, act, int(normalize), mode,
                                             _ptr(mean), _ptr(istd), _ptr(gamma), _ptr(sums),
                                             ctypes.c_double(count), _ptr(dh_out), _ptr(dh_colsum_out), _ptr(ws),
                                             self._stream()), 'cgc_bn_act_l2_bwd')

    def sage_narrow_bwd(self, dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, agg, lda, fin, weight,
                        dagg_out, dwdb_out):
        if F > 32 or fin > 32:
            return False
        self._dev(dy, hn, rinv, mean, istd, gamma, sums, agg, weight, dagg_out, dwdb_out)
        ws = torch.empty(int(self.lib.cgc_sage_narrow_ws_floats(n, fin, F)), dtype=torch.float32, device=hn.device)
        self._chk(self.lib.cgc_sage_narrow_bwd(_ptr(dy), ldy, _ptr(hn), _ptr(rinv), n, F, act, int(normalize), mode, _ptr(mean),
                                               _ptr(istd), _ptr(gamma), _ptr(sums), ctypes.c_double(count), _ptr(agg), lda, fin,
                                               _ptr(weight), _ptr(dagg_out), _ptr(dwdb_out), _ptr(ws), self._stream()),
                  'cgc_sage_narrow_bwd')
        return True

    def colsum(self, x, ld, n, F, out):
        self._dev(x, out)
        nblk = self.lib.cgc_stats_blocks(n, F)
        ws = torch.empty(max(nblk, 1) * 2 * F, dtype=torch.float32, device=x.device)
        self._chk(self.lib.cgc_colsum(_ptr(x), ld, n, F, _ptr(out), _ptr(ws), self._stream()), 'cgc_colsum')

    # -- softmax / readout
    def softmax_fwd(self, x, n, C, out, ld=None):
        self._dev(x, out)
        self._chk(self.lib.cgc_softmax_fwd(_ptr(x), n, C, C if ld is None else ld, _ptr(out), self._stream()), 'cgc_softmax_fwd')

    def softmax_bwd(self, S, dS, n, C, dx_out, dx_colsum_out=None, ld=None):
        self._dev(S, dS, dx_out, dx_colsum_out)
        ws = self._slots(n, C, S.device) if dx_colsum_out is not None else None
        self._chk(self.lib.cgc_softmax_bwd(_ptr(S), _ptr(dS), n, C, C if ld is None else ld, _ptr(dx_out), _ptr(dx_colsum_out), _ptr(ws),
                                           self._stream()), 'cgc_softmax_bwd')

    def segment_max_fwd(self, x, gptr, B, D, nmax, out, arg_out):
        self._dev(x, gptr, out, arg_out)
        self._chk(self.lib.cgc_segment_max_fwd(_ptr(x), _ptr(gptr), B, D, nmax, _ptr(out), _ptr(arg_out),
                                               self._stream()), 'cgc_segment_max_fwd')

    def segment_max_bwd(self, dout, arg, B, D, dx_zeroed):
        self._dev(dout, arg, dx_zeroed)
        self._chk(self.lib.cgc_segment_max_bwd(_ptr(dout), _ptr(arg), B, D, _ptr(dx_zeroed), self._stream()),
                  'cgc_segment_max_bwd')

    def segment_max_bwd_full(self, dout, arg, gptr, B, D, nmax, dx_out):
        self._dev(dout, arg, gptr, dx_out)
        self._chk(self.lib.cgc_segment_max_bwd_full(_ptr(dout), _ptr(arg), _ptr(gptr), B, D, nmax, _ptr(dx_out), self._stream()),
                  'cgc_segment_max_bwd_full')

    # -- jumping knowledge
    def jk_supported(self, C):
        return bool(self.lib.cgc_jk_supported(int(C)))

    @staticmethod
    def _ptr_array(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def jk_fwd(self, xs, n, npad, C, lstm, w_att, b_att, out, HS, CS):
        self._dev(xs, w_att, b_att, out, HS, CS, *lstm)
        self._chk(self.lib.cgc_jk_lstm_fwd(_ptr(xs), n, npad, C, self._ptr_array(lstm), _ptr(w_att), _ptr(b_att), _ptr(out),
                                           _ptr(HS), _ptr(CS), self._stream()), 'cgc_jk_lstm_fwd')

    def jk_bwd(self, xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, DGT, INT, DHC):
        self._dev(xs, dout, w_att, b_att, HS, CS, dxs, DGT, INT, DHC, *lstm)
        self._chk(self.lib.cgc_jk_lstm_bwd(_ptr(xs), _ptr(dout), n, npad, C, self._ptr_array(lstm), _ptr(w_att), _ptr(b_att),
                                           _ptr(HS), _ptr(CS), _ptr(dxs), _ptr(DGT), _ptr(INT), _ptr(DHC), self._stream()),
                  'cgc_jk_lstm_bwd')

    def jk_bwd_params(self, xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, G_out):
        self._dev(xs, dout, w_att, b_att, HS, CS, dxs, G_out, *lstm)
        H = 3 * C // 2
        ng, ni, ktot = 4 * H + 1, C + 2 * H + 1, 3 * npad
        ws = torch.empty(int(self.lib.cgc_jk_bwd_ws_floats(C)), dtype=torch.float32, device=xs.device)
        rc = self.lib.cgc_jk_lstm_bwd_params(_ptr(xs), _ptr(dout), n, npad, C, self._ptr_array(lstm), _ptr(w_att), _ptr(b_att),
                                             _ptr(HS), _ptr(CS), _ptr(dxs), _ptr(G_out), _ptr(ws), self._stream())
        if rc == 0:
            return
        if rc != -1:
            self._chk(rc, 'cgc_jk_lstm_bwd_params')
        # unaligned buffers / other channel counts: staged path -- gate gradients and cell inputs transposed, one batched NT
        # GEMM per direction over K slices of 768 columns, combined deterministically
        dev = xs.device
        DGT = torch.empty(2, ng, ktot, dtype=torch.float32, device=dev)
        INT = torch.empty(2, ni, ktot, dtype=torch.float32, device=dev)
        DHC = torch.empty(2, 2, H, npad, dtype=torch.float32, device=dev)
        self.jk_bwd(xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, DGT, INT, DHC)
        kp = 768
        parts = ktot // kp
        ws2 = torch.empty(parts, ng * ni, dtype=torch.float32, device=dev)
        for d in range(2):
            self.gemm(DGT[d], INT[d], ws2, ng, ni, kp, False, True, ktot, ktot, ni, 1.0, 0.0, None, parts, kp, kp, ng * ni)
            self.reduce_batch_sum(ws2, G_out[d], parts, ng * ni, 0.0)

    def jk_unpack_param_grads(self, G, C):
        self._dev(G)
        flat = torch.empty(int(self.lib.cgc_jk_param_grad_floats(int(C))), dtype=torch.float32, device=G.device)
        self._chk(self.lib.cgc_jk_unpack_param_grads(_ptr(G), int(C), _ptr(flat), self._stream()), 'cgc_jk_unpack_param_grads')
        return split_jk_param_grads(flat, C)

    # -- dense adjacency ops
    def dense_rownorm_fwd(self, A, R, C, out, invd_out, ge1_out):
        self._dev(A, out, invd_out, ge1_out)
        self._chk(self.lib.cgc_dense_rownorm_fwd(_ptr(A), R, C, _ptr(out), _ptr(invd_out), _ptr(ge1_out),
                                                 self._stream()), 'cgc_dense_rownorm_fwd')

    def dense_rownorm_bwd(self, dOut, Anorm, invd, ge1, R, C, dA_out):
        self._dev(dOut, Anorm, invd, ge1, dA_out)
        self._chk(self.lib.cgc_dense_rownorm_bwd(_ptr(dOut), _ptr(Anorm), _ptr(invd), _ptr(ge1), R, C, _ptr(dA_out),
                                                 self._stream()), 'cgc_dense_rownorm_bwd')

    def dense_renorm_fwd(self, A, R, C, p, out):
        self._dev(A, out)
        self._chk(self.lib.cgc_dense_renorm_fwd(_ptr(A), R, C, ctypes.c_float(p), _ptr(out), self._stream()),
                  'cgc_dense_renorm_fwd')

    def dense_renorm_bwd(self, A, dOut, R, C, p, dA_out):
        self._dev(A, dOut, dA_out)
        self._chk(self.lib.cgc_dense_renorm_bwd(_ptr(A), _ptr(dOut), R, C, ctypes.c_float(p), _ptr(dA_out),
                                                self._stream()), 'cgc_dense_renorm_bwd')

    def adj_prep_fwd(self, A, R, C, p, At_out, An_out, invd_out, ge1_out):
        self._dev(A, At_out, An_out, invd_out, ge1_out)
        self._chk(self.lib.cgc_adj_prep_fwd(_ptr(A), R, C, ctypes.c_float(-1.0 if p is None else p), _ptr(At_out), _ptr(An_out),
                                            _ptr(invd_out), _ptr(ge1_out), self._stream()), 'cgc_adj_prep_fwd')

    def adj_prep_bwd(self, A, An, invd, ge1, gAn, gAt, R, C, p, dA_out):
        self._dev(A, An, invd, ge1, gAn, gAt, dA_out)
        self._chk(self.lib.cgc_adj_prep_bwd(_ptr(A), _ptr(An), _ptr(invd), _ptr(ge1), _ptr(gAn), _ptr(gAt), R, C,
                                            ctypes.c_float(-1.0 if p is None else p), _ptr(dA_out), self._stream()), 'cgc_adj_prep_bwd')
