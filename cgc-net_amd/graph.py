"""Device-side structure of one batch of cell graphs at level 1 (replaces the reference's dense
``[B, Nmax, Nmax]`` adjacency, model/utils.py:3-36 / model/network.py:237-243).

Layout in HBM (all int32 / fp32, contiguous):
  rowptr[n+1], col[cap]            CSR, row = aggregating centre, columns sorted, duplicates collapsed
  val[cap] or None                 per-edge weight (only after ``_re_norm_adj``; None = all ones)
  inv_d[n]                         1 / max(rowsum, 1)   (DenseSAGEConv's clamped mean divisor)
  t_rowptr[n+1], t_col, t_perm     the transpose (for backward); t_perm -> slot in ``val``; t_val = val in transposed order
  gptr[B+1]                        first node of each graph; nmax = max nodes per graph, npad = rows per graph of the dense layout (host ints)
n = total real nodes: padding rows of the dense layout are never materialised, their effect on
BatchNorm statistics and on the max readout is applied analytically (count = B*npad).
"""
import torch

from . import kernels


class BatchGraph(object):
    def __init__(self, n, counts, device, npad=None, gptr=None):
        self.n = int(n)
        self.counts = [int(c) for c in counts]
        self.B = len(self.counts)
        self.nmax = max(self.counts) if self.counts else 0
        # rows per graph of the reference's dense layout: the largest graph for a sparse Batch (model/utils.py:21), the
        # loader's fixed padding (dataflow/data.py:234,268) for the dense-tuple input form
        self.npad = self.nmax if npad is None else int(npad)
        assert self.npad >= self.nmax, 'dense padding smaller than the largest graph'
        ptr = [0]
        for c in self.counts:
            ptr.append(ptr[-1] + c)
        assert ptr[-1] == self.n, 'batch vector and x disagree on the node count'
        self.gptr_host = ptr
        # the offsets on the device: the Batch's own copy when the collate made one (it came over with the batch); building it here
        # is a blocking host-to-device copy that also waits for everything queued before it -- the whole previous step
        if (torch.is_tensor(gptr) and gptr.dtype == torch.int32 and gptr.device == torch.device(device)
                and gptr.numel() == self.B + 1 and gptr.is_contiguous()):
            self.gptr = gptr
        else:
            self.gptr = torch.tensor(ptr, dtype=torch.int32, device=device)
        # visiting sequence for the wide aggregation when the graphs are LARGE (thousands of nodes: the tensors no longer fit the
        # Infinity Cache, so the recency hints are worthless, and four graphs of unequal size per XCD leave up to 20 % imbalance):
        # graphs sorted by size and dealt to the 8 XCDs in serpentine order, each XCD's graphs listed consecutively
        self.gorder = None
        if self.B >= 8 and self.B % 8 == 0 and self.nmax >= 4000:
            by_size = sorted(range(self.B), key=lambda b: -self.counts[b])
            lanes = [[] for _ in range(8)]
            for q, b in enumerate(by_size):
                r, x = divmod(q, 8)
                lanes[x if r % 2 == 0 else 7 - x].append(b)
            # (a pinned staging buffer + non_blocking copy: torch.tensor(list, device=...) is a blocking copy that also waits for
            # everything queued before it -- the whole previous step)
            host = torch.tensor([b for lane in lanes for b in lane], dtype=torch.int32)
            if torch.device(device).type == 'cuda':
                self.gorder = host.pin_memory().to(device, non_blocking=True)
            else:
                self.gorder = host.to(device)
        self.val = None
        self.t_val = None
        self.renorm_p = None
        self.spatial = False

    @property
    def padded_rows(self):
        """B * Nmax: the row count the reference's BatchNorm sees at level 1."""
        return self.B * self.npad

    @staticmethod
    def node_counts_of(batch):
        """Per-graph node counts, from host metadata when the Batch carries it (no device sync)."""
        counts = getattr(batch, '_node_counts', None)
        if counts is not None:
            return list(counts)
        b = batch.batch
        num = int(b[-1]) + 1 if b.numel() else 0            # model/utils.py:17
        return torch.bincount(b, minlength=num).tolist()    # one D2H sync for foreign Batch objects

    @classmethod
    def from_batch(cls, batch, renorm_p=None):
        x, edge_index = batch.x, batch.edge_index
        g = cls(x.shape[0], cls.node_counts_of(batch), x.device, getattr(batch, '_dense_rows', None), getattr(batch, '_gptr', None))
        # the nodes of every graph are listed grid cell by grid cell (data.spatial_order; the Batch says so).  What makes the wide
        # aggregation faster on such a batch is the node ORDER itself (the gather's re-reads hit nearer caches); the note only
        # travels on as cgc_spmm_graphs' visit bit 2, which the default gather kernel ignores (it selects the experimental LDS-staged
        # kernel under CGC_SPMM_PATCH=1).  Never about results.
        g.spatial = bool(getattr(batch, '_spatial', False))
        # the collate's note that the edge list is grouped by graph (Batch._eptr), if it still describes this edge_index
        eptr = getattr(batch, '_eptr', None)
        if not (torch.is_tensor(eptr) and eptr.dtype == torch.int32 and eptr.device == edge_index.device and eptr.numel() == g.B + 1
                and getattr(batch, '_etotal', -1) == edge_index.shape[1]):
            eptr = None
        g._build(edge_index.contiguous(), renorm_p, eptr, int(getattr(batch, '_emax', 0)))
        return g

    def _build(self, edge_index, renorm_p, eptr=None, emax=0):
        K = kernels.get()
        if hasattr(K, 'graph_build'):          # the library's composite entry point: one call, two allocations
            s = K.graph_build(edge_index, self.n, None if renorm_p is None else float(renorm_p),
                              gptr=self.gptr if eptr is not None else None, eptr=eptr, num_graphs=self.B, nmax=self.nmax, emax=emax)
            self.rowptr, self.col, self.rowidx = s['rowptr'], s['col'], s['rowidx']
            self.t_rowptr, self.t_col, self.t_perm = s['t_rowptr'], s['t_col'], s['t_perm']
            self.cap, self.bad_edges = s['cap'], s['bad_edges']
            self.val, self.t_val, self.inv_d = s['val'], s['t_val'], s['inv_d']
            self.renorm_p = None if renorm_p is None else float(renorm_p)
            return
        s = K.csr_build(edge_index, self.n, add_diag=renorm_p is not None)
        self.rowptr, self.col, self.rowidx = s['rowptr'], s['col'], s['rowidx']
        self.t_rowptr, self.t_col, self.t_perm = s['t_rowptr'], s['t_col'], s['t_perm']
        self.cap = s['cap']
        self.bad_edges = s.get('bad_edges')        # device counter of edges with ids outside [0, n) (dropped by the build)
        if renorm_p is not None:
            self.renorm_p = float(renorm_p)
            self.val = torch.empty(max(self.cap, 1), dtype=torch.float32, device=self.col.device)
            K.edge_renorm(self.rowptr, self.col, self.n, self.renorm_p, self.val)
            # the same weights in transposed slot order: the backward aggregations then need no t_perm indirection (a
            # dependent scalar load per edge in the gather kernels; 12 us of the wide transposed SpMM)
            self.t_val = torch.empty_like(self.val)
            K.csr_transpose_vals(self.t_rowptr, self.t_perm, self.val, self.n, self.t_val)
        self.inv_d = torch.empty(max(self.n, 1), dtype=torch.float32, device=self.col.device)
        K.csr_invdeg(self.rowptr, self.val, self.n, self.inv_d)

    def validate(self):
        """Raise if edge_index referred to nodes outside the batch (one device sync; network.py calls it when
        CGC_VALIDATE_INPUTS=1, tests always).  The reference fails with an IndexError in to_dense_adj (model/utils.py:28-33)."""
        bad = int(self.bad_edges) if self.bad_edges is not None else 0
        if bad:
            raise IndexError('%d edge(s) reference node ids outside [0, %d): edge_index does not belong to this batch' % (bad, self.n))
        return self

    @property
    def nnz(self):
        return int(self.rowptr[self.n])     # device sync; only for reporting


def uniform_ptr(B, C, device, _cache={}):
    """gptr of B graphs with exactly C nodes each (levels 2 and 3)."""
    key = (B, C, str(device))
    if key not in _cache:
        _cache[key] = torch.arange(0, (B + 1) * C, C, dtype=torch.int32, device=device)
    return _cache[key]
