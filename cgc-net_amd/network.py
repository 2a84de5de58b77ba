"""nn.Module surface of the hot path -- same class names, constructor signatures, forward contracts
and state_dict keys as the reference's ``model/network.py`` (so that ``train.py:178-183`` and a
reference checkpoint work unchanged), with the arithmetic scheduled onto the HIP kernels (ops.py).

What differs from the reference, by design:
* level 1 never builds the dense ``[B, Nmax, Nmax]`` adjacency (model/network.py:237-243): the batch is
  kept flat (``[Ntot, F]`` rows + CSR, graph.py); masks are implicit; the effect of the zero padding rows
  on BatchNorm statistics (model/network.py:101-107) and on the max readout (:264) is applied analytically.
* ``(S^T A) S`` is evaluated as ``S^T (A S)`` with ``A S`` a sparse product (model/network.py:207).
* no ``.cuda()`` calls inside the model: it runs on the device its parameters live on.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels, native, ops
from .graph import BatchGraph, uniform_ptr

EPS = 1e-15
_VALIDATE_INPUTS = os.environ.get('CGC_VALIDATE_INPUTS', '0') == '1'     # one device sync per batch: off by default
RENORM_P = 0.4      # model/network.py:260,271,280
REORDER_MIN_NODES = 4000    # graphs at least this large get their nodes listed grid cell by grid cell (_spatially_ordered)


def default_gemm_mode():
    """An encoder's ``gemm_mode`` when nothing else sets it: ``CGC_GEMM_16BIT`` = ``0`` / ``exact`` (kernels.GEMM_EXACT), ``1`` / ``bf16``
    (GEMM_SPLIT_BF16), ``2`` / ``f16`` (GEMM_SPLIT_F16); unset: the older switch ``CGC_GEMM_SPLIT_BF16`` (0 / 1 / 2); neither set: 2."""
    v = os.environ.get('CGC_GEMM_16BIT')
    if v is None:
        v = os.environ.get('CGC_GEMM_SPLIT_BF16', '2')
    names = {'0': 0, 'exact': 0, 'off': 0, '1': 1, 'bf16': 1, '2': 2, 'f16': 2, 'fp16': 2}
    if v.lower() not in names:
        raise ValueError('CGC_GEMM_16BIT / CGC_GEMM_SPLIT_BF16 = %r: expected one of %s' % (v, sorted(names)))
    return names[v.lower()]


def _activation_module(name):
    assert name in ('relu', 'elu', 'leakyrelu')          # model/network.py:84-91
    return {'relu': nn.ReLU, 'elu': nn.ELU, 'leakyrelu': nn.LeakyReLU}[name](inplace=True)


# ------------------------------------------------------------------------------------------------
# operators (torch_geometric 1.2.1 signatures)
# ------------------------------------------------------------------------------------------------
class DenseSAGEConv(nn.Module):
    """``DenseSAGEConv(in, out, normalize=True, bias=True)``; weight is [in, out] as in PyG."""

    def __init__(self, in_channels, out_channels, normalize=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.normalize = in_channels, out_channels, normalize
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def project(self, agg):
        """[rows, in] aggregated features -> [rows, out]: linear (+bias); the caller applies l2/act/BN."""
        return ops.linear_bias(agg, self.weight, self.bias, out_in_layout=False)

    def forward(self, x, adj, mask=None, add_loop=True):
        x = x.unsqueeze(0) if x.dim() == 2 else x
        adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
        B, N, _ = x.shape
        if add_loop:
            adj = adj.clone()
            idx = torch.arange(N, device=adj.device)
            adj[:, idx, idx] = 1
        agg = ops.bmatmul(ops.rownorm_clamp(adj), x)
        out = self.project(agg.reshape(B * N, -1))
        if self.normalize:
            out = ops.l2_act_bn(out, None, B * N, 'identity', True, self.training)
        out = out.view(B, N, -1)
        if mask is not None:
            out = out * mask.view(B, N, 1).to(out.dtype)
        return out

    def __repr__(self):
        return '%s(%d, %d)' % (self.__class__.__name__, self.in_channels, self.out_channels)


class DenseGINConv(nn.Module):
    """``DenseGINConv(nn, eps=0)``: nn(adj @ x [+ (1+eps) x]); the MLP's Linear layers run on the HIP GEMM."""

    def __init__(self, nn_module, eps=0.0):
        super().__init__()
        self.nn = nn_module
        self.register_buffer('eps', torch.tensor([float(eps)]))

    def mlp(self, h):
        for layer in self.nn:
            if isinstance(layer, nn.Linear):
                h = ops.linear_bias(h, layer.weight, layer.bias, out_in_layout=True)
            elif isinstance(layer, nn.ReLU):
                h = ops.l2_act_bn(h, None, h.shape[0], 'relu', False, self.training)
            elif isinstance(layer, nn.ELU):
                h = ops.l2_act_bn(h, None, h.shape[0], 'elu', False, self.training)
            elif isinstance(layer, nn.LeakyReLU):
                h = ops.l2_act_bn(h, None, h.shape[0], 'leakyrelu', False, self.training)
            else:
                raise NotImplementedError(type(layer))
        return h

    def forward(self, x, adj, mask=None, add_loop=True):
        x = x.unsqueeze(0) if x.dim() == 2 else x
        adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
        B, N, _ = x.shape
        out = ops.bmatmul(adj, x)
        if add_loop:
            out = (1 + float(self.eps)) * x + out
        out = self.mlp(out.reshape(B * N, -1)).view(B, N, -1)
        if mask is not None:
            out = out * mask.view(B, N, 1).to(out.dtype)
        return out


# ------------------------------------------------------------------------------------------------
class DenseJK(nn.Module):
    """LSTM-attention jumping knowledge over a block's three layer outputs (model/network.py:11-55).
    Three layers with an even channel count <= 32 run on the fused HIP kernels (csrc/jk.hip, csrc/jk_mfma.hip); anything else on
    torch.nn.LSTM."""

    def __init__(self, mode, channels=None, num_layers=None):
        super().__init__()
        self.channel = channels
        self.mode = mode.lower()
        assert self.mode in ['cat', 'max', 'lstm']
        if self.mode == 'lstm':
            assert channels is not None and num_layers is not None
            self.lstm = nn.LSTM(channels, channels * num_layers // 2, bidirectional=True, batch_first=True)
            self.att = nn.Linear(2 * channels * num_layers // 2, 1)
        self.reset_parameters()

    def reset_parameters(self):
        if hasattr(self, 'lstm'):
            self.lstm.reset_parameters()
        if hasattr(self, 'att'):
            self.att.reset_parameters()

    def forward(self, xs):
        """[..., layers*channels] -> [..., channels]; works on [B, N, 3C] and on flat [Ntot, 3C] rows alike."""
        lead = xs.shape[:-1]
        layers = xs.shape[-1] // self.channel
        if layers == 3 and ops.K().jk_supported(self.channel):
            # fused bi-LSTM + attention kernels (one thread per node); nn.LSTM / nn.Linear only hold the parameters
            return ops.dense_jk(xs.reshape(-1, 3 * self.channel), self.lstm, self.att).reshape(*lead, self.channel)
        getattr(ops.K(), '_dev', lambda *a: None)(xs)    # GPU tensors only here as well (no silent CPU path through torch)
        seq = xs.reshape(-1, layers, self.channel)   # [rows, layers, channels]: other shapes stay on torch.nn.LSTM (MIOpen)
        alpha, _ = self.lstm(seq)
        alpha = torch.softmax(self.att(alpha).squeeze(-1), dim=-1)
        return (seq * alpha.unsqueeze(-1)).sum(dim=1).reshape(*lead, self.channel)

    def __repr__(self):
        return '{}({})'.format(self.__class__.__name__, self.mode)


# ------------------------------------------------------------------------------------------------
class GNN_Module(nn.Module):
    """Three convolutions, each followed by activation THEN BatchNorm, concatenated; optional Linear
    (model/network.py:57-125)."""

    def __init__(self, input_dim, hidden_dim, embedding_dim, bias, bn, add_loop, lin=True, gcn_name='SAGE',
                 sync=False, activation='relu', jk=False):
        super().__init__()
        if sync:
            raise NotImplementedError('sync BatchNorm: the reference path creates only bn1 and cannot run '
                                      '(SURVEY Appendix C.4)')
        self.jk, self.add_loop, self.gcn_name, self.activation = jk, add_loop, gcn_name, activation
        self.gcn1 = self._gcn(gcn_name, input_dim, hidden_dim, bias, activation)
        self.active1 = _activation_module(activation)
        self.gcn2 = self._gcn(gcn_name, hidden_dim, hidden_dim, bias, activation)
        self.active2 = _activation_module(activation)
        self.gcn3 = self._gcn(gcn_name, hidden_dim, embedding_dim, bias, activation)
        self.active3 = _activation_module(activation)
        self.use_bn = bool(bn)
        if bn:
            self.bn1 = nn.BatchNorm1d(hidden_dim)
            self.bn2 = nn.BatchNorm1d(hidden_dim)
            self.bn3 = nn.BatchNorm1d(embedding_dim)
        self.lin = nn.Linear(2 * hidden_dim + embedding_dim, embedding_dim) if lin is True else None

    @staticmethod
    def _gcn(name, input_dim, hidden_dim, bias, activation='relu'):
        if name == 'SAGE':
            return DenseSAGEConv(input_dim, hidden_dim, normalize=True, bias=bias)
        nn1 = nn.Sequential(nn.Linear(input_dim, hidden_dim), _activation_module(activation),
                            nn.Linear(hidden_dim, hidden_dim))
        return DenseGINConv(nn1)

    @property
    def mean_aggregation(self):
        return self.gcn_name == 'SAGE'

    # -- core: rows are nodes; `aggregate` maps [rows, F] -> [rows, F]; `count` = rows the BatchNorm of the
    #    dense layout would see (B*Nmax); `row_mask` only for the dense API with padded rows.
    def run_rows(self, x, aggregate, count, agg0=None, row_mask=None, softmax=False):
        """``softmax=True`` (pool blocks): return softmax(lin(cat)) -- the assignment matrix -- instead of the logits."""
        outs, h = [], x
        for k in (1, 2, 3):
            conv = getattr(self, 'gcn%d' % k)
            bn = getattr(self, 'bn%d' % k) if self.use_bn else None
            agg = agg0 if (k == 1 and agg0 is not None) else aggregate(h)
            training = bn.training if bn is not None else self.training     # (nn.BatchNorm1d follows its OWN flag: model/network.py:101-107)
            if self.mean_aggregation and row_mask is None:
                h = ops.sage_project(agg, conv.weight, conv.bias, bn, count, self.activation, conv.normalize, training)
                outs.append(h)
                continue
            if self.mean_aggregation:
                z, normalize = conv.project(agg), conv.normalize
            else:
                if self.add_loop:
                    agg = agg + (1 + float(conv.eps)) * h
                z, normalize = conv.mlp(agg), False
            if row_mask is None:
                h = ops.l2_act_bn(z, bn, count, self.activation, normalize, training)
            else:   # padded dense layout: conv output is masked BEFORE activation/BN (model/network.py:114)
                if normalize:
                    z = ops.l2_act_bn(z, None, count, 'identity', True, training)
                h = ops.l2_act_bn(z * row_mask, bn, count, self.activation, False, training)
            outs.append(h)
        if row_mask is None:
            return self._tail(outs, softmax)
        h = torch.cat(outs, dim=-1)
        if row_mask is not None:
            h = h * row_mask
        if self.lin is not None:
            h = ops.linear_bias(h, self.lin.weight, self.lin.bias, out_in_layout=True) * row_mask
        return ops.softmax_rows(h) if softmax else h

    def _tail(self, outs, softmax=False):
        """cat[x1,x2,x3] (-> Linear (-> softmax)) of the three layer outputs, unpadded rows."""
        if self.lin is not None:
            # Linear over cat[x1,x2,x3] without the concatenation: the two narrow pieces are joined (cheap), the wide one
            # ([rows, cluster count]) is the main operand of the same GEMM
            return ops.linear_cat([torch.cat(outs[:2], dim=-1), outs[2]], self.lin.weight, self.lin.bias, softmax)
        h = torch.cat(outs, dim=-1)
        return ops.softmax_rows(h) if softmax else h

    def forward_graph(self, x, g, agg0=None, softmax=False):
        """Level-1 path on flat rows + CSR (``g``: graph.BatchGraph).  Returns [Ntot, width]."""
        mean = self.mean_aggregation
        return self.run_rows(x, lambda h: ops.aggregate(h, g, mean), g.padded_rows, agg0, None, softmax)

    def forward(self, x, adj, mask=None):
        """Dense-tensor contract of the reference: x [B,N,F], adj [B,N,N], mask [B,N,1] or None."""
        B, N, _ = x.shape
        if self.add_loop and self.mean_aggregation:
            adj = adj.clone()
            idx = torch.arange(N, device=adj.device)
            adj[:, idx, idx] = 1
        a = ops.rownorm_clamp(adj) if self.mean_aggregation else adj
        row_mask = mask.reshape(B * N, 1).to(x.dtype) if mask is not None else None
        out = self.run_rows(x.reshape(B * N, -1), lambda h: ops.bmatmul(a, h.view(B, N, -1)).view(B * N, -1),
                            B * N, None, row_mask)
        return out.view(B, N, -1)


def run_blocks_paired(emb, pool, x, aggregate, count, agg0):
    """The embedding block and the assignment block of one level, layer by layer.  Layer k of both blocks aggregates over
    the SAME adjacency (model/network.py:258-262: ``GCN_embed_k(x, adj)`` and ``GCN_pool_k(x, adj)``), so the two
    aggregations run as ONE pass over it, ``A [h_embed | h_pool]`` (the first layer's inputs are identical: ``agg0``).  At
    level 2 that pass reads the dense 166 MB adjacency, at level 1 it is one gather instead of two.  Returns the blocks' layer
    outputs (lists of three)."""
    he = hp = x
    pair = None
    outs_e, outs_p = [], []
    for k in (1, 2, 3):
        ce, cp = getattr(emb, 'gcn%d' % k), getattr(pool, 'gcn%d' % k)
        be = getattr(emb, 'bn%d' % k) if emb.use_bn else None
        bp = getattr(pool, 'bn%d' % k) if pool.use_bn else None
        if k == 1:
            ae = ap = agg0
        else:
            we = he.shape[1]
            agg = aggregate(ops.join_cols(he, hp, pair) if pair is not None else torch.cat([he, hp], dim=-1))
            ae, ap = ops.split_cols(agg, we)
        pair = None
        if k < 3 and ce.out_channels + cp.out_channels <= 256:
            # layers whose outputs are aggregated together next: both blocks write into ONE [rows, we + wp] buffer
            pair = torch.empty(x.shape[0], ce.out_channels + cp.out_channels, dtype=torch.float32, device=x.device)
        he = ops.sage_project(ae, ce.weight, ce.bias, be, count, emb.activation, ce.normalize, be.training if be is not None else emb.training,
                              out=None if pair is None else (pair, 0))
        hp = ops.sage_project(ap, cp.weight, cp.bias, bp, count, pool.activation, cp.normalize, bp.training if bp is not None else pool.training,
                              out=None if pair is None else (pair, ce.out_channels))
        outs_e.append(he)
        outs_p.append(hp)
    return outs_e, outs_p


def _pairable(emb, pool):
    return emb.mean_aggregation and pool.mean_aggregation and not emb.add_loop and not pool.add_loop


# ------------------------------------------------------------------------------------------------
class SoftPoolingGcnEncoder(nn.Module):
    """model/network.py:127-291.  ``forward(data)`` takes a Batch-like object (``.x .edge_index .batch .y``)
    when ``load_data_sparse`` else the tuple ``(x[B,N,F], adj[B,N,N], num_nodes[B][, label])``; returns
    ``(logits, loss)`` in training mode and ``logits`` in eval mode."""

    def __init__(self, max_num_nodes, input_dim, hidden_dim, embedding_dim, bias, bn, assign_hidden_dim, label_dim,
                 assign_ratio=0.25, pred_hidden_dims=[50], concat=True, gcn_name='SAGE',
                 collect_assign=False, load_data_sparse=False, norm_adj=False,
                 activation='relu', drop_out=0., jk=False):
        super().__init__()
        self.jk, self.drop_out, self.norm_adj = jk, drop_out, norm_adj
        self.load_data_sparse, self.collect_assign = load_data_sparse, collect_assign
        self.assign_matrix = []
        kw = dict(add_loop=False, gcn_name=gcn_name, activation=activation, jk=jk)
        assign_dim = int(max_num_nodes * assign_ratio)
        self.GCN_embed_1 = GNN_Module(input_dim, hidden_dim, embedding_dim, bias, bn, lin=False, **kw)
        if jk:
            self.jk1 = DenseJK('lstm', hidden_dim, 3)
        self.GCN_pool_1 = GNN_Module(input_dim, assign_hidden_dim, assign_dim, bias, bn, **kw)
        if concat and not jk:
            input_dim = hidden_dim * 2 + embedding_dim
        else:
            input_dim = embedding_dim
        assign_dim = int(assign_dim * assign_ratio)
        self.GCN_embed_2 = GNN_Module(input_dim, hidden_dim, embedding_dim, bias, bn, lin=False, **kw)
        if jk:
            self.jk2 = DenseJK('lstm', hidden_dim, 3)
        self.GCN_pool_2 = GNN_Module(input_dim, assign_hidden_dim, assign_dim, bias, bn, **kw)
        self.GCN_embed_3 = GNN_Module(input_dim, hidden_dim, embedding_dim, bias, bn, lin=False, **kw)
        if jk:
            self.jk3 = DenseJK('lstm', hidden_dim, 3)
        self.pred_model = self.build_readout_module(input_dim * 3, pred_hidden_dims, label_dim, activation)
        self.last_graph = None
        # levels run through the step sequencer (native.py / csrc/exec.hip: one library call per level and direction) whenever it
        # covers the configuration; False (or CGC_NATIVE=0): always the per-operator path (ops.py), one autograd node per operator
        self.native = os.environ.get('CGC_NATIVE', '1') != '0'
        self.native_head = os.environ.get('CGC_NATIVE_HEAD', '1') != '0'     # classification head + loss as one kernel each way
        self.reorder_large = os.environ.get('CGC_REORDER', '1') != '0'       # see _spatially_ordered
        # How the products on the 128 x 128 route -- from ~450 output tiles up: the six dominant products of a step (assignment Linear,
        # S^T(AS), their backward: model/network.py:121-122, 206-207) -- are computed (include/cgc_hip.h: cgc_gemm_f32_ws):
        #   2 = kernels.GEMM_SPLIT_F16 (the module's default): three fp16 MFMA pairs of operands scaled per output tile
        #       (csrc/gemm_half.hip); those products 1.75x faster than exact, the step 1.37x;
        #   1 = kernels.GEMM_SPLIT_BF16: six bf16 MFMA pairs (csrc/gemm_split.hip; no scaling, no range caveat); 1.4x / 1.2x;
        #   0 = kernels.GEMM_EXACT: the fp32 matrix-core chain for every product -- what bench.py's headline `value` is measured with.
        # All three give the reference's results to fp32 rounding (tests/test_split_gemm_gpu.py, tests/test_half_gemm_gpu.py have the
        # bounds; the reference's fixtures run through each with every product forced onto that route).  Smaller products are exact
        # in every mode.  CGC_GEMM_16BIT = 0 | bf16 | f16 chooses (default_gemm_mode above).
        self.gemm_mode = default_gemm_mode()
        self._unorder = None

    def __getstate__(self):
        """copy.deepcopy / pickle (EMA or best-model snapshots, torch.save(model)): the sequencer's per-encoder caches hold ctypes
        structs with raw device pointers and views of one step's buffers -- they stay behind and are rebuilt on first use."""
        state = self.__dict__.copy()
        for k in native.TRANSIENT + ('last_graph',):
            state.pop(k, None)
        state['last_graph'] = None
        return state

    def build_readout_module(self, pred_input_dim, pred_hidden_dims, label_dim, activation):
        if len(pred_hidden_dims) == 0:
            return nn.Linear(pred_input_dim, label_dim)
        layers = []
        for pred_dim in pred_hidden_dims:
            layers += [nn.Linear(pred_input_dim, pred_dim), _activation_module(activation)]
            pred_input_dim = pred_dim
            if self.drop_out > 0:
                layers.append(nn.Dropout(self.drop_out))
        layers.append(nn.Linear(pred_dim, label_dim))
        return nn.Sequential(*layers)

    def _head(self, readouts):
        """``pred_model(cat(readouts))`` (model/network.py:286-287) with its Linear layers on the library's own GEMM: the first one
        takes the three readouts as K segments (no concatenation), activations / dropout stay the registered modules.  [B, 60]
        -> 50 -> 3 is all launch overhead: the hipBLASLt path behind nn.Linear costs ~45 us of host time per call."""
        layers = list(self.pred_model) if isinstance(self.pred_model, nn.Sequential) else [self.pred_model]
        h = None
        for i, m in enumerate(layers):
            if isinstance(m, nn.Linear):
                if i == 0:
                    h = ops.linear_cat(readouts, m.weight, m.bias)
                else:
                    h = ops.linear_bias(h, m.weight, m.bias, out_in_layout=True)
            else:
                h = m(h)
        return h

    # -- inputs --------------------------------------------------------------------------------
    class _Flat(object):
        pass

    def _flat_from_dense(self, x, adj, num_nodes):
        """The tuple input form (model/network.py:253-256): 0/1 dense adjacency -> flat rows + edge list."""
        counts = [int(c) for c in (num_nodes.tolist() if torch.is_tensor(num_nodes) else num_nodes)]
        B, N, _ = adj.shape
        dev = x.device
        cnt = torch.tensor(counts, device=dev)
        off = torch.cumsum(cnt, 0) - cnt
        real = torch.arange(N, device=dev).unsqueeze(0) < cnt.unsqueeze(1)         # [B, N]
        b, r, c = (adj * real.unsqueeze(2).to(adj.dtype)).nonzero(as_tuple=True)   # padded rows carry no edges
        flat = self._Flat()
        flat.x = x[real]
        flat.edge_index = torch.stack([off[b] + r, off[b] + c])
        flat._node_counts = counts
        flat._dense_rows = N          # the loader's padding (dataflow/data.py:234,268): BN row count B*N, readout vs zero rows
        return flat

    # -- stages --------------------------------------------------------------------------------
    def _use_native(self, x):
        """Training steps (autograd on) and inference under no_grad both run on the sequencer; anything in between -- eval mode with
        gradients enabled, training mode under no_grad -- stays on the per-operator path."""
        if not (self.native and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad):
            return False
        return torch.is_grad_enabled() if self.training else not torch.is_grad_enabled()

    def _native_level(self, level, x, adj, g=None):
        """One level through the sequencer; None when it does not cover this configuration."""
        emb = getattr(self, 'GCN_embed_%d' % level)
        pool = getattr(self, 'GCN_pool_%d' % level) if level < 3 else None
        jk = getattr(self, 'jk%d' % level) if self.jk else None
        fin = x.shape[-1]
        prep = native.prepared(self, level, emb, pool, jk, fin)
        if prep is None:
            return None
        if level == 1:
            desc = native.describe_from(prep, 1, g.B, g.n, 0, g.nmax, g.npad, g.padded_rows)
            gptr = g.gptr
        else:
            B, Cn, _ = x.shape
            desc = native.describe_from(prep, level, B, B * Cn, Cn, 0, 0, B * Cn)
            gptr = uniform_ptr(B, Cn, x.device)
            x, adj = ops._f32c(x).view(B * Cn, fin), ops._f32c(adj)
        if desc is None:
            return None
        assign = [] if (self.collect_assign and pool is not None) else None
        if self.training:
            out = native.level(self, desc, emb, pool, jk, g, gptr, x.contiguous(), adj, assign, prep)
        else:
            out = native.level_eval(self, desc, emb, pool, jk, g, gptr, x.contiguous(), adj, assign, prep)
        if assign:
            s = assign[0]
            self.assign_matrix.append(self._pad_assign(s, g) if level == 1 else s.view(desc.B, desc.rows_per_graph, -1))
        return out

    def _spatially_ordered(self, data):
        """Large graphs (thousands of nodes): list the nodes of every graph grid cell by grid cell before the CSR is built.  The
        network is invariant to the node order (tested at full size); the order decides how far apart in HBM the rows are that
        the wide aggregation A*S gathers together -- with > ~4000 nodes per graph an arbitrary order (the reference's is the
        sampler's pick order, dataflow/data.py:210-219) makes the working set of a (graph, column tile) exceed an XCD's 4 MiB
        L2: 0.30 of the HBM peak instead of 0.40 (DESIGN.md, K4 at C5).  Returns (data', inverse permutation) or (data, None)."""
        pos, bvec = getattr(data, 'pos', None), getattr(data, 'batch', None)
        counts = getattr(data, '_node_counts', None)
        if (not self.reorder_large or pos is None or bvec is None or counts is None or not data.x.is_cuda
                or max(counts, default=0) < REORDER_MIN_NODES or pos.shape[0] != data.x.shape[0]):
            return data, None
        cell = 100.0                                                      # the k-NN radius (dataflow/data.py:348)
        p = pos.detach().to(torch.float32)
        cx, cy = torch.floor(p[:, 0] / cell).long(), torch.floor(p[:, 1] / cell).long()
        cx, cy = cx - cx.min(), cy - cy.min()
        nx, ny = int(cx.max()) + 1, int(cy.max()) + 1                     # (two host reads per batch; only taken for large graphs)
        by_x = torch.sort(p[:, 0], stable=True)[1]
        key = ((bvec.long() * ny + cy) * nx + cx)[by_x]                   # graph, grid row, grid column; x breaks ties inside a cell
        perm = by_x[torch.sort(key, stable=True)[1]]
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        flat = self._Flat()
        flat.x = data.x[perm]
        flat.edge_index = inv[data.edge_index]
        flat._node_counts = counts
        flat._spatial = True
        for k in ('_gptr', '_dense_rows'):
            if getattr(data, k, None) is not None:
                setattr(flat, k, getattr(data, k))
        return flat, inv

    def _level1(self, data):
        data, self._unorder = self._spatially_ordered(data)
        g = BatchGraph.from_batch(data, RENORM_P if self.norm_adj else None)
        if _VALIDATE_INPUTS:
            g.validate()
        self.last_graph = g
        x = data.x
        if self._use_native(x):
            out = self._native_level(1, x, None, g)
            if out is not None:
                return out
        emb_blk, pool_blk = self.GCN_embed_1, self.GCN_pool_1
        agg0 = ops.aggregate(x, g, emb_blk.mean_aggregation)     # shared by both blocks' first conv
        outs_p = None
        if _pairable(emb_blk, pool_blk):
            outs_e, outs_p = run_blocks_paired(emb_blk, pool_blk, x, lambda h: ops.aggregate(h, g, True), g.padded_rows, agg0)
            embed = emb_blk._tail(outs_e)
        else:
            embed = emb_blk.forward_graph(x, g, agg0)
        if self.jk:
            embed = self.jk1(embed)
        readout = ops.segment_max(embed, g.gptr, g.B, g.npad)
        # the assignment matrix last: the wide aggregation A*S right behind it finds S's tail in the Infinity Cache
        s = pool_blk._tail(outs_p, softmax=True) if outs_p is not None else pool_blk.forward_graph(x, g, agg0, softmax=True)
        if self.collect_assign:
            self.assign_matrix.append(self._pad_assign(s.detach(), g))
        xn, an = ops.diff_pool_sparse(embed, s, g)
        return readout, xn, an

    def _pad_assign(self, s, g):
        """[Ntot, C] -> the reference's [B, Nmax, C]; its padded rows hold softmax(0) = 1/C.  Rows return to the caller's node order."""
        if getattr(self, '_unorder', None) is not None:
            s = s[self._unorder]
        out = s.new_full((g.B, g.npad, s.shape[1]), 1.0 / s.shape[1])
        for b in range(g.B):
            out[b, :g.counts[b]] = s[g.gptr_host[b]:g.gptr_host[b + 1]]
        return out

    def _dense_level(self, level, x, adj):
        B, C, _ = x.shape
        if self._use_native(x.detach()):
            out = self._native_level(level, x, adj)
            if out is not None:
                return out
        emb_blk = getattr(self, 'GCN_embed_%d' % level)
        if emb_blk.mean_aggregation:       # re-normalisation + clamped row normalisation: one fused pass each way
            adj, a = ops.adj_prep(adj, RENORM_P if self.norm_adj else None)
        else:
            if self.norm_adj:
                adj = ops.renorm_dense(adj, RENORM_P)
            a = adj
        shared = ops.SharedGrad() if (torch.is_grad_enabled() and a.requires_grad) else None   # one d(adjacency) buffer per level

        def aggregate(h):
            return ops.bmatmul(a, h.view(B, C, -1), shared=shared).view(B * C, -1)
        xf = x.reshape(B * C, -1)
        agg0 = aggregate(xf)
        pool_blk = getattr(self, 'GCN_pool_%d' % level) if level < 3 else None
        outs_p = None
        if pool_blk is not None and _pairable(emb_blk, pool_blk):
            outs_e, outs_p = run_blocks_paired(emb_blk, pool_blk, xf, aggregate, B * C, agg0)
            embed = emb_blk._tail(outs_e)
        else:
            embed = emb_blk.run_rows(xf, aggregate, B * C, agg0)
        if self.jk:
            embed = getattr(self, 'jk%d' % level)(embed)
        readout = ops.segment_max(embed, uniform_ptr(B, C, x.device), B, C)
        if level == 3:
            return readout, None, None
        s = pool_blk._tail(outs_p, softmax=True) if outs_p is not None else pool_blk.run_rows(xf, aggregate, B * C, agg0, None, True)
        if self.collect_assign:
            self.assign_matrix.append(s.detach().view(B, C, -1))
        xn, an = ops.diff_pool_dense(embed.view(B, C, -1), adj, s.view(B, C, -1))
        return readout, xn, an

    def _dense_levels(self, x, adj):
        out2, x, adj = self._dense_level(2, x, adj)
        out3, _, _ = self._dense_level(3, x, adj)
        return out2, out3

    def forward(self, data):
        if not kernels.is_native():
            return self._forward(data)
        # the per-operator path reads the GEMM mode from the kernel table at call time: installed for this forward only (every
        # autograd node remembers the mode it ran under for its backward: ops._bind_gemm_mode), restored so that it does not leak
        # into whatever uses the table next
        K = kernels.get()
        keep, K.gemm_mode = K.gemm_mode, int(getattr(self, 'gemm_mode', 0))
        try:
            return self._forward(data)
        finally:
            K.gemm_mode = keep

    def _forward(self, data):
        self.assign_matrix = []
        if self.load_data_sparse:
            label = data.y
        else:
            label = data[3] if self.training else None
            data = self._flat_from_dense(data[0], data[1], data[2])
        out1, x, adj = self._level1(data)
        out2, out3 = self._dense_levels(x, adj)
        if self.native_head and self.training and torch.is_grad_enabled() and out1.is_cuda and label.dtype == torch.int64:
            # head + mean cross-entropy as one kernel each way (native.head; csrc/head.hip)
            res = native.head(self.pred_model, [out1, out2, out3], label, True, owner=self)
            if res is not None:
                return res
        output = self._head([out1, out2, out3])
        if self.training:
            cls_loss = F.cross_entropy(output, label.view(-1))
            return output, cls_loss
        return output
