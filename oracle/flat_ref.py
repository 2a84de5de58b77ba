"""Op-level checker for the HIP kernels: the KernelSpec contract in plain torch (CPU or any device).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each method restates, on the flat / CSR layout the
kernels use, the arithmetic the reference performs on padded dense tensors; the reference lines are
cited in cgc-net_amd/kernels.py::KernelSpec next to each entry.  tests/ use it in two ways:

* GPU: every ``HipKernels`` method is compared with the method of the same name here;
* CPU: monkeypatched in place of the kernel table, it lets the product's autograd layer and modules
  (ops.py, network.py) run end to end so that the flat formulation -- including the hand-derived
  backward passes -- is checked against oracle/dense_ref.py and the golden fixtures without a GPU.

It is never selected by the product.
"""
import torch
import torch.nn.functional as F

import cgc_net_amd  # noqa: F401
from cgc_net_amd.kernels import KernelSpec, L2_EPS, RENORM_EPS


def _act(x, code):
    if code == 0:
        return x
    if code == 1:
        return torch.relu(x)
    if code == 2:
        return F.elu(x)
    if code == 3:
        return F.leaky_relu(x, 0.01)
    raise ValueError(code)


def _dact(x, code):
    if code == 0:
        return torch.ones_like(x)
    if code == 1:
        return (x > 0).to(x.dtype)
    if code == 2:
        return torch.where(x > 0, torch.ones_like(x), torch.exp(x))
    if code == 3:
        return torch.where(x > 0, torch.ones_like(x), torch.full_like(x, 0.01))
    raise ValueError(code)


def _mat(t, rows, cols, ld, offset=0):
    """[rows, cols] window with leading dimension ld starting ``offset`` elements after t's first element."""
    return t.as_strided((rows, cols), (ld, 1), t.storage_offset() + offset)


class TorchKernels(KernelSpec):
    # ------------------------------------------------------------------ graph structure
    def csr_build(self, edge_index, n, add_diag):
        row, col = edge_index[0], edge_index[1]
        ok = (row >= 0) & (row < n) & (col >= 0) & (col < n)
        bad = int((~ok).sum())
        row, col = row[ok], col[ok]
        if add_diag:
            ar = torch.arange(n, dtype=torch.int64, device=row.device)
            row, col = torch.cat([row, ar]), torch.cat([col, ar])
        cap = row.numel()
        key = torch.unique(row * n + col)            # sorted + de-duplicated
        row, col = key // n, key % n
        nnz = key.numel()
        i32 = torch.int32

        def ptr_of(idx):
            return torch.cat([idx.new_zeros(1), torch.bincount(idx, minlength=n).cumsum(0)]).to(i32)

        def padded(v):
            out = torch.zeros(max(cap, 1), dtype=i32, device=v.device)
            out[:nnz] = v.to(i32)
            return out
        tkey, tperm = torch.sort(col * n + row)      # transposed order: by col, then by row
        return {'rowptr': ptr_of(row), 'col': padded(col), 'rowidx': padded(row),
                't_rowptr': ptr_of(col), 't_col': padded(tkey % n), 't_perm': padded(tperm), 'cap': cap + bad,
                'bad_edges': torch.tensor([bad], dtype=i32, device=row.device)}

    @staticmethod
    def _rows(rowptr, n):
        cnt = (rowptr[1:] - rowptr[:-1]).long()
        return torch.repeat_interleave(torch.arange(n, device=rowptr.device), cnt), int(rowptr[n])

    def collate(self, x, mean, std, gptr, num_graphs, batch_out, edge_index, eptr):
        """dataflow/data.py:353 (z-score) + torch_geometric Batch.from_data_list (batch vector, edge offsets)."""
        if mean is not None:
            x.copy_((x - mean) / std)
        gp = gptr.long()
        counts = gp[1:] - gp[:-1]
        if batch_out is not None:
            batch_out.copy_(torch.repeat_interleave(torch.arange(num_graphs), counts))
        if edge_index is not None:
            ep = eptr.long()
            edge_index += torch.repeat_interleave(gp[:-1], ep[1:] - ep[:-1]).unsqueeze(0)

    def farthest_point_sample(self, pos, gptr, num_graphs, max_nodes, start, optr, out, table16=False):
        """common/utils.py:187-197 with the distance-table rows replaced by squared coordinate distances (float64), or
        (table16) by the table's own entries: float32 arithmetic + astype(int16) as in
        dataflow/construct_feature_graph.py:17-24."""
        import numpy as np
        p = pos.detach().cpu().numpy().astype(np.float32 if table16 else np.float64)
        gp, op, st = gptr.cpu().numpy(), optr.cpu().numpy(), start.cpu().numpy()
        res = np.zeros(int(op[-1]), dtype=np.int32)
        for g in range(num_graphs):
            lo, hi, k = int(gp[g]), int(gp[g + 1]), int(op[g + 1] - op[g])
            if hi <= lo or k <= 0:
                continue
            q = p[lo:hi]
            cur = min(max(int(st[g]), 0), hi - lo - 1)
            dist = np.full(hi - lo, np.inf)
            for i in range(k):
                res[op[g] + i] = lo + cur
                dx, dy = q[:, 0] - q[cur, 0], q[:, 1] - q[cur, 1]
                d = np.sqrt(dx ** 2 + dy ** 2).astype(np.int16) if table16 else dx * dx + dy * dy
                dist = np.minimum(dist, d)
                cur = int(dist.argmax())
        out.copy_(torch.from_numpy(res).to(out.device))

    def radius_knn(self, pos, gptr, num_graphs, r, k, loop):
        """cKDTree.query(k+1, distance_upper_bound=r+1e-8) per graph (torch_cluster 1.4.2's CPU radius_graph, SURVEY B.5;
        call site dataflow/data.py:348), graph by graph with global node ids; neighbours by (distance, index)."""
        import numpy as np
        from scipy.spatial import cKDTree
        p = pos.detach().cpu().numpy().astype(np.float64)
        gp = gptr.cpu().numpy()
        rows, cols = [], []
        for g in range(num_graphs):
            lo, hi = int(gp[g]), int(gp[g + 1])
            if hi <= lo:
                continue
            q = p[lo:hi]
            kk = min(k + 1, hi - lo)
            d, c = cKDTree(q).query(q, k=kk, distance_upper_bound=r + 1e-8)
            d, c = d.reshape(hi - lo, kk), c.reshape(hi - lo, kk)
            for i in range(hi - lo):
                cand = [(d[i, u], int(c[i, u])) for u in range(kk) if c[i, u] < hi - lo and (loop or c[i, u] != i)]
                cand.sort()
                rows += [lo + i] * len(cand)
                cols += [lo + j for _, j in cand]
        return torch.tensor([rows, cols], dtype=torch.int64).reshape(2, -1)

    def edge_renorm(self, rowptr, col, n, p, val_out):
        rows, nnz = self._rows(rowptr, n)
        c = col[:nnz].long()
        off = (c != rows).to(torch.float32)
        cnt = torch.zeros(n, device=col.device).index_add_(0, rows, off)
        w = (1.0 / (cnt + RENORM_EPS)) * (1 - p)
        val_out[:nnz] = torch.where(c == rows, torch.full_like(off, p), w[rows])

    def csr_transpose_vals(self, t_rowptr, t_perm, val, n, t_val_out):
        nnz = int(t_rowptr[n])
        t_val_out[:nnz] = val[t_perm[:nnz].long()]

    def csr_invdeg(self, rowptr, val, n, out):
        rows, nnz = self._rows(rowptr, n)
        v = val[:nnz] if val is not None else torch.ones(nnz, device=rowptr.device)
        s = torch.zeros(n, device=rowptr.device).index_add_(0, rows, v)
        out.copy_(1.0 / s.clamp(min=1))

    def spmm(self, rowptr, col, perm, val, pre, post, x, out, n, width, gptr=None, num_graphs=0, nmax=0, visit=0, ld=None, gorder=None):
        rows, nnz = self._rows(rowptr, n)
        c = col[:nnz].long()
        w = torch.ones(nnz, device=x.device)
        if val is not None:
            w = val[perm[:nnz].long()] if perm is not None else val[:nnz]
        if pre is not None:
            w = w * pre[c]
        acc = torch.zeros(n, width, device=x.device).index_add_(0, rows, w.unsqueeze(1) * x[c])
        if post is not None:
            acc = acc * post.unsqueeze(1)
        out.copy_(acc)

    # ------------------------------------------------------------------ dense contractions
    def gemm(self, A, B, C, M, N, K, transA, transB, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None,
             batch=1, strideA=0, strideB=0, strideC=0, gptr=None, ragged=0, max_ragged=0, ragged_total=0, extra=()):
        if ragged == 1:
            assert not transA
        if ragged == 2:
            assert transA and not transB
        g = gptr.tolist() if gptr is not None else None
        for b in range(batch):
            m, k = M, K
            oa, ob, oc = b * strideA, b * strideB, b * strideC
            if ragged == 1:
                m = g[b + 1] - g[b]
                assert m <= max_ragged
                oa += g[b] * lda
                oc += g[b] * ldc
            elif ragged == 2:
                k = g[b + 1] - g[b]
                assert k <= max_ragged
                oa += g[b] * lda
                ob += g[b] * ldb
            elif ragged == 3:        # uniform row chunks of max_ragged rows; K = rows per outer item (include/cgc_hip.h)
                assert transA and not transB and not extra
                parts = -(-K // max_ragged)
                outer, part = divmod(b, parts)
                g0 = outer * K + part * max_ragged
                k = min(max_ragged, K - part * max_ragged)
                oa, ob = g0 * lda, g0 * ldb
            a = _mat(A, k, m, lda, oa).t() if transA else _mat(A, m, k, lda, oa)
            bm = _mat(B, N, k, ldb, ob).t() if transB else _mat(B, k, N, ldb, ob)
            c = _mat(C, m, N, ldc, oc)
            prod = a @ bm
            for (Ax, Bx, ldax, ldbx, Kx, sAx, sBx) in extra:       # concatenated-K product, segment by segment
                assert ragged != 2
                oax = b * sAx + (g[b] * ldax if ragged == 1 else 0)
                ax = _mat(Ax, Kx, m, ldax, oax).t() if transA else _mat(Ax, m, Kx, ldax, oax)
                bx = _mat(Bx, N, Kx, ldbx, b * sBx).t() if transB else _mat(Bx, Kx, N, ldbx, b * sBx)
                prod = prod + ax @ bx
            r = alpha * prod
            if beta != 0.0:
                r = r + beta * c
            if bias is not None:
                r = r + bias
            c.copy_(r)

    def reduce_batch_sum(self, ws, out, parts, numel, beta=0.0):
        s = ws.reshape(-1)[:parts * numel].view(parts, numel).sum(0)
        o = out.view(-1)
        o.copy_(s + beta * o if beta != 0.0 else s)

    def reduce_batched(self, ws, out, outer, parts, numel, beta=0.0):
        s = ws.reshape(-1)[:outer * parts * numel].view(outer, parts, numel).sum(1)
        o = out.view(outer, numel)
        o.copy_(s + beta * o if beta != 0.0 else s)

    # ------------------------------------------------------------------ conv epilogue
    def l2norm_act_stats(self, h, n, F_, normalize, act, hn_out, rinv_out, stats_out):
        if normalize:
            r = 1.0 / h.norm(dim=1).clamp(min=L2_EPS)
        else:
            r = torch.ones(n, device=h.device)
        hn = h * r.unsqueeze(1)
        hn_out.copy_(hn)
        rinv_out.copy_(r)
        if stats_out is not None:
            o = _act(hn, act)
            stats_out[0] = o.double().sum(0)
            stats_out[1] = (o.double() * o.double()).sum(0)

    def l2norm_act_bn(self, h, n, F_, normalize, act, hn_out, rinv_out, count, eps, momentum, running_mean, running_var,
                      num_batches_tracked, mean_out, istd_out):
        stats = torch.zeros(2, F_, dtype=torch.float64, device=h.device)
        self.l2norm_act_stats(h, n, F_, normalize, act, hn_out, rinv_out, stats)
        self.bn_finalize(stats, count, eps, momentum, running_mean, running_var, mean_out, istd_out)
        if num_batches_tracked is not None:
            num_batches_tracked += 1

    def bn_finalize(self, stats, count, eps, momentum, running_mean, running_var, mean_out, istd_out):
        mean = stats[0].double() / count
        var = (stats[1].double() / count - mean * mean).clamp(min=0)
        mean_out.copy_(mean.float())
        istd_out.copy_((1.0 / torch.sqrt(var + eps)).float())
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * (var * count / max(count - 1, 1)).float())

    def bn_act_apply(self, hn, n, F_, act, mean, istd, gamma, beta, y_out, ldy):
        o = _act(hn, act)
        if mean is not None:
            o = (o - mean) * istd * gamma + beta
        _mat(y_out, n, F_, ldy).copy_(o)

    def bn_bwd_reduce(self, dy, ldy, hn, n, F_, act, mean, istd, sums_out):
        d = _mat(dy, n, F_, ldy)
        xhat = (_act(hn, act) - mean) * istd
        sums_out[0] = d.sum(0)
        sums_out[1] = (d * xhat).sum(0)

    def bn_act_l2_bwd(self, dy, ldy, hn, rinv, n, F_, act, normalize, mode, mean, istd, gamma, sums, count, dh_out,
                      dh_colsum_out=None):
        d = _mat(dy, n, F_, ldy)
        if mode == 2:
            xhat = (_act(hn, act) - mean) * istd
            do = gamma * istd * (d - sums[0] / count - xhat * sums[1] / count)
        elif mode == 1:
            do = gamma * istd * d
        else:
            do = d
        dhn = do * _dact(hn, act)
        if normalize:
            dot = (hn * dhn).sum(1, keepdim=True)
            r = rinv.unsqueeze(1)
            dh = torch.where(r < 1.0 / L2_EPS, r * (dhn - hn * dot), dhn / L2_EPS)
        else:
            dh = dhn
        dh_out.copy_(dh)
        if dh_colsum_out is not None:
            dh_colsum_out.copy_(dh.sum(0))

    def colsum(self, x, ld, n, F_, out):
        out.copy_(_mat(x, n, F_, ld).sum(0))

    # ------------------------------------------------------------------ softmax / readout
    def softmax_fwd(self, x, n, C, out, ld=None):      # (padded rows arrive as strided views: nothing to do with ld here)
        out.copy_(torch.softmax(x, dim=1))

    def softmax_bwd(self, S, dS, n, C, dx_out, dx_colsum_out=None, ld=None):
        dx_out.copy_(S * (dS - (dS * S).sum(1, keepdim=True)))
        if dx_colsum_out is not None:
            dx_colsum_out.copy_(dx_out.sum(0))

    def segment_max_fwd(self, x, gptr, B, D, nmax, out, arg_out):
        g = gptr.tolist()
        for b in range(B):
            lo, hi = g[b], g[b + 1]
            if hi > lo:
                v, a = x[lo:hi].max(dim=0)      # first index on ties (CPU semantics of torch.max)
                a = a + lo
            else:
                v = torch.full((D,), float('-inf'), device=x.device)
                a = torch.full((D,), -1, dtype=torch.long, device=x.device)
            if hi - lo < nmax:
                pad_wins = v < 0
                v = torch.where(pad_wins, torch.zeros_like(v), v)
                a = torch.where(pad_wins, torch.full_like(a, -1), a)
            out[b] = v
            arg_out[b] = a.to(arg_out.dtype)

    def segment_max_bwd(self, dout, arg, B, D, dx_zeroed):
        a = arg.long()
        cols = torch.arange(D, device=dout.device).expand(B, D)
        ok = a >= 0
        dx_zeroed[a[ok], cols[ok]] = dout[ok]

    # ------------------------------------------------------------------ jumping-knowledge attention
    def jk_supported(self, C):
        return C % 2 == 0

    @staticmethod
    def _jk_math(xs, C, lstm, w_att, b_att):
        """bi-LSTM (torch gate order i,f,g,o) over the 3 layer embeddings + attention; returns out, hs, cs, gates."""
        H = 3 * C // 2
        x = xs.reshape(-1, 3, C)
        hs, cs = {}, {}
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = lstm[4 * d:4 * d + 4]
            h = x.new_zeros(x.shape[0], H)
            c = x.new_zeros(x.shape[0], H)
            for s in range(3):
                t = s if d == 0 else 2 - s
                g = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
                i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
                c = f * c + i * gg
                h = o * torch.tanh(c)
                hs[(d, t)], cs[(d, t)] = h, c
        score = torch.stack([torch.cat([hs[(0, t)], hs[(1, t)]], 1) @ w_att + b_att for t in range(3)], 1)
        a = torch.softmax(score, dim=1)
        return (x * a.unsqueeze(-1)).sum(1), hs, cs

    def jk_fwd(self, xs, n, npad, C, lstm, w_att, b_att, out, HS, CS):
        H = 3 * C // 2
        o, hs, cs = self._jk_math(xs.detach(), C, [p.detach() for p in lstm], w_att.detach().reshape(-1), b_att.detach())
        out.copy_(o)
        for d in range(2):
            for t in range(3):
                HS[(d * 3 + t) * H:(d * 3 + t + 1) * H, :n] = hs[(d, t)].t()
                CS[(d * 3 + t) * H:(d * 3 + t + 1) * H, :n] = cs[(d, t)].t()

    def jk_bwd(self, xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, DGT, INT, DHC):
        """Same CONTRACT as the HIP kernel (the products DGT[d] @ INT[d]^T hold the parameter gradients), obtained
        here from autograd: the gate-gradient rows are reconstructed so that the product matches."""
        H = 3 * C // 2
        with torch.enable_grad():
            x = xs.detach().clone().requires_grad_()
            ps = [p.detach().clone().requires_grad_() for p in lstm]
            wa, ba = w_att.detach().clone().reshape(-1).requires_grad_(), b_att.detach().clone().requires_grad_()
            o, hs, cs = self._jk_math(x, C, ps, wa, ba)
            grads = torch.autograd.grad(o, [x] + ps + [wa, ba], dout)
        dxs.copy_(grads[0])
        # encode the parameter gradients in the (DGT, INT) factorisation: put G_d in the first columns against an identity
        DGT.zero_()
        INT.zero_()
        ni = C + 2 * H + 1
        for d in range(2):
            g = torch.zeros(4 * H + 1, ni)
            g[:4 * H, :C] = grads[1 + 4 * d]
            g[:4 * H, C:C + H] = grads[2 + 4 * d]
            g[:4 * H, C + H] = grads[3 + 4 * d]
            g[4 * H, C + H + 1:] = grads[9][d * H:(d + 1) * H]
            g[4 * H, C + H] = grads[10].reshape(()) if d == 0 else 0.0
            assert 3 * npad >= ni, 'twin needs 3*npad >= C+2H+1 columns'
            DGT[d][:, :ni] = g.to(DGT.device)
            INT[d][:, :ni] = torch.eye(ni, device=INT.device)


    def jk_bwd_params(self, xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, G_out):
        H = 3 * C // 2
        ng, ni, ktot = 4 * H + 1, C + 2 * H + 1, 3 * npad
        DGT, INT = torch.zeros(2, ng, ktot, device=xs.device), torch.zeros(2, ni, ktot, device=xs.device)
        DHC = torch.zeros(2, 2, H, npad, device=xs.device)
        self.jk_bwd(xs, dout, n, npad, C, lstm, w_att, b_att, HS, CS, dxs, DGT, INT, DHC)
        for d in range(2):
            G_out[d].copy_(DGT[d] @ INT[d].t())

    # ------------------------------------------------------------------ dense adjacency ops
    def dense_rownorm_fwd(self, A, R, C, out, invd_out, ge1_out):
        a = A.reshape(R, C)
        s = a.sum(1)
        d = s.clamp(min=1)
        out.view(R, C).copy_(a / d.unsqueeze(1))
        invd_out.copy_(1.0 / d)
        ge1_out.copy_((s >= 1).float())

    def dense_rownorm_bwd(self, dOut, Anorm, invd, ge1, R, C, dA_out):
        g, an = dOut.reshape(R, C), Anorm.reshape(R, C)
        t = (g * an).sum(1)
        dA_out.view(R, C).copy_(invd.unsqueeze(1) * (g - (ge1 * t).unsqueeze(1)))

    @staticmethod
    def _offdiag(R, C, device):
        m = torch.ones(R, C, device=device)
        m[torch.arange(R, device=device), torch.arange(R, device=device) % C] = 0
        return m

    def dense_renorm_fwd(self, A, R, C, p, out):
        a = A.reshape(R, C)
        m = self._offdiag(R, C, a.device)
        a0 = a * m
        q = 1.0 / (a0.sum(1, keepdim=True) + RENORM_EPS)
        out.view(R, C).copy_(a0 * q * (1 - p) + (1 - m) * p)

    def dense_renorm_bwd(self, A, dOut, R, C, p, dA_out):
        a, g = A.reshape(R, C), dOut.reshape(R, C)
        m = self._offdiag(R, C, a.device)
        q = 1.0 / ((a * m).sum(1, keepdim=True) + RENORM_EPS)
        t = (g * a * m).sum(1, keepdim=True)
        dA_out.view(R, C).copy_((1 - p) * q * (g - q * t) * m)

    def adj_prep_fwd(self, A, R, C, p, At_out, An_out, invd_out, ge1_out):
        src = A
        if p is not None:
            self.dense_renorm_fwd(A, R, C, p, At_out)
            src = At_out
        self.dense_rownorm_fwd(src, R, C, An_out, invd_out, ge1_out)

    def adj_prep_bwd(self, A, An, invd, ge1, gAn, gAt, R, C, p, dA_out):
        dAt = torch.empty_like(An)
        self.dense_rownorm_bwd(gAn, An, invd, ge1, R, C, dAt)
        if gAt is not None:
            dAt = dAt + gAt.reshape(dAt.shape)
        if p is None:
            dA_out.copy_(dAt.reshape(dA_out.shape))
        else:
            self.dense_renorm_bwd(A, dAt, R, C, p, dA_out)

    def jk_unpack_param_grads(self, G, C):
        from cgc_net_amd.kernels import split_jk_param_grads
        H = 3 * C // 2
        parts = []
        for d in range(2):
            parts += [G[d, :4 * H, :C].reshape(-1), G[d, :4 * H, C:C + H].reshape(-1), G[d, :4 * H, C + H], G[d, :4 * H, C + H]]
        parts += [G[0, 4 * H, C + H + 1:], G[1, 4 * H, C + H + 1:], G[0, 4 * H, C + H].reshape(1)]
        return split_jk_param_grads(torch.cat([p.reshape(-1) for p in parts]), C)

    def sage_wide_fwd(self, agg, lda, weight, bias, n, Kin, F, normalize, act, hn_out, rinv_out, stats, count, eps, momentum,
                      running_mean, running_var, num_batches_tracked, mean_out, istd_out):
        if Kin > 32 or F > 1664:
            return False
        h = agg[:, :Kin] @ weight
        if bias is not None:
            h = h + bias
        h = h.contiguous()
        if stats:
            self.l2norm_act_bn(h, n, F, normalize, act, hn_out, rinv_out, count, eps, momentum, running_mean, running_var,
                               num_batches_tracked, mean_out, istd_out)
        else:
            self.l2norm_act_stats(h, n, F, normalize, act, hn_out, rinv_out, None)
        return True

    def segment_max_bwd_full(self, dout, arg, gptr, B, D, nmax, dx_out):
        gp = gptr.long()
        dx_out[int(gp[0]):int(gp[B])] = 0
        self.segment_max_bwd(dout, arg, B, D, dx_out)

    def sage_narrow_bwd(self, dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, agg, lda, fin, weight,
                        dagg_out, dwdb_out):
        if F > 32 or fin > 32:
            return False
        dh = torch.empty(n, F, dtype=hn.dtype, device=hn.device)
        db = torch.empty(F, dtype=hn.dtype, device=hn.device)
        self.bn_act_l2_bwd(dy, ldy, hn, rinv, n, F, act, normalize, mode, mean, istd, gamma, sums, count, dh, db)
        a = agg[:, :fin]
        if dagg_out is not None:
            dagg_out.copy_(dh @ weight.t())
        dwdb_out[:fin * F].copy_((a.t() @ dh).reshape(-1))
        dwdb_out[fin * F:].copy_(db)
        return True
