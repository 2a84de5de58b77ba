"""Dense CPU restatement of the CGC-Net hot path (oracle; see oracle/__init__.py).

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  The restatement is functional: parameters live in small
holder modules whose attribute names reproduce the reference ``state_dict`` keys
 so that a reference checkpoint / a golden fixture loads with
``load_state_dict`` unchanged.

PyG 1.2.1 semantics (DenseSAGEConv, DenseGINConv, to_dense_batch) are restated
from the published package -- parity is UNPINNED at that boundary.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

EPS = 1e-15  # model/network.py:8


# ----------------------------------------------------------------------------
# graph -> dense tensors
# ----------------------------------------------------------------------------
def node_counts(batch, num_graphs=None):
    """Nodes per graph.  model/utils.py:17-19 (scatter_('add', ones, batch))."""
    if num_graphs is None:
        num_graphs = int(batch[-1]) + 1  # model/utils.py:17: batch sorted, last graph non-empty
    return torch.bincount(batch, minlength=num_graphs)


def to_dense_adj(edge_index, batch):
    """COO edge list -> [B, Nmax, Nmax] 0/1 float adjacency.

    model/utils.py:15-36.  Assignment semantics: a repeated edge still gives 1,
    no symmetrisation, row = aggregating centre, col = neighbour.
    """
    counts = node_counts(batch)
    start = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    nmax = int(counts.max())
    adj = torch.zeros(counts.numel(), nmax, nmax, dtype=torch.float32)
    src_graph = batch[edge_index[0]]
    r = edge_index[0] - start[src_graph]
    c = edge_index[1] - start[batch[edge_index[1]]]
    adj[src_graph, r, c] = 1.0
    return adj


def to_dense_batch(x, batch):
    """Flat node features -> ([B, Nmax, F] zero padded, counts[B]).

    PyG 1.2.1 ``to_dense_batch`` returns the per-graph node COUNTS as its second
    value (the reference iterates it as counts, model/network.py:175-179,242).
    """
    counts = node_counts(batch)
    start = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    nmax = int(counts.max())
    dense = x.new_zeros(counts.numel(), nmax, x.shape[-1])
    local = torch.arange(x.shape[0]) - start[batch]
    dense[batch, local] = x
    return dense, counts


def node_mask(nmax, counts):
    """[B, Nmax, 1] float mask of real rows.  model/network.py:172-180."""
    ar = torch.arange(nmax).unsqueeze(0)
    return (ar < counts.view(-1, 1)).to(torch.float32).unsqueeze(-1)


# ----------------------------------------------------------------------------
# operators (PyG 1.2.1 restated -- unpinned)
# ----------------------------------------------------------------------------
def dense_sage(x, adj, weight, bias=None, mask=None, add_loop=True, normalize=True):
    """DenseSAGEConv.forward of torch-geometric 1.2.1 (call sites model/network.py:95,114-116).

    mean aggregation with the divisor clamped at 1, ONE weight (no root weight),
    optional L2 row normalisation, optional row mask.
    """
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if adj.dim() == 2:
        adj = adj.unsqueeze(0)
    if add_loop:
        adj = adj.clone()
        i = torch.arange(adj.shape[1])
        adj[:, i, i] = 1.0
    out = torch.matmul(adj, x)
    out = out / adj.sum(dim=-1, keepdim=True).clamp(min=1)
    out = torch.matmul(out, weight)
    if bias is not None:
        out = out + bias
    if normalize:
        out = F.normalize(out, p=2, dim=-1)
    if mask is not None:
        out = out * mask.view(x.shape[0], x.shape[1], 1).to(x.dtype)
    return out


def dense_gin(x, adj, mlp, mask=None, add_loop=True, eps=0.0):
    """DenseGINConv.forward of torch-geometric 1.2.1 (model/network.py:97-99)."""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if adj.dim() == 2:
        adj = adj.unsqueeze(0)
    out = torch.matmul(adj, x)
    if add_loop:
        out = (1 + eps) * x + out
    out = mlp(out)
    if mask is not None:
        out = out * mask.view(x.shape[0], x.shape[1], 1).to(x.dtype)
    return out


class DenseSAGEConv(nn.Module):
    """Parameter holder with PyG's layout: weight [in, out], bias [out]; U(-1/sqrt(in), 1/sqrt(in))."""

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

    def forward(self, x, adj, mask=None, add_loop=True):
        return dense_sage(x, adj, self.weight, self.bias, mask, add_loop, self.normalize)


class DenseGINConv(nn.Module):
    def __init__(self, nn_module, eps=0.0):
        super().__init__()
        self.nn = nn_module
        self.register_buffer('eps', torch.tensor([float(eps)]))  # PyG 1.2.1: non-trainable eps buffer

    def forward(self, x, adj, mask=None, add_loop=True):
        return dense_gin(x, adj, self.nn, mask, add_loop, float(self.eps))


# ----------------------------------------------------------------------------
# the reference's own arithmetic (pinned by tests/golden)
# ----------------------------------------------------------------------------
def make_activation(name):
    """model/network.py:84-91."""
    return {'relu': nn.ReLU, 'elu': nn.ELU, 'leakyrelu': nn.LeakyReLU}[name]()


def re_norm_adj(adj, p, mask=None):
    """model/network.py:183-191.  Zero diagonal, row-normalise to (1-p), diagonal := p, row mask.

    The reference zeroes the diagonal of its input IN PLACE (Appendix C.6); the
    oracle clones first when the input needs grad elsewhere -- same values.
    """
    n = adj.shape[1]
    i = torch.arange(n)
    adj = adj.clone()
    adj[:, i, i] = 0
    out = adj / (adj.sum(-1, keepdim=True) + EPS) * (1 - p)
    out[:, i, i] = p
    if mask is not None:
        out = out * mask
    return out


def diff_pool(x, adj, s, mask=None):
    """model/network.py:194-208.  Returns (S^T X, (S^T A) S, S_after_softmax)."""
    s = torch.softmax(s, dim=-1)
    s_soft = s
    if mask is not None:
        s = s * mask
    st = s.transpose(1, 2)
    return torch.matmul(st, x), torch.matmul(torch.matmul(st, adj), s), s_soft


class GNNBlock(nn.Module):
    """model/network.py:57-125 (GNN_Module): three convs, ACT before BN, concat, optional Linear."""

    def __init__(self, input_dim, hidden_dim, embedding_dim, bias=True, bn=True, add_loop=False,
                 lin=True, gcn_name='SAGE', activation='relu'):
        super().__init__()
        self.add_loop = add_loop
        dims = [(input_dim, hidden_dim), (hidden_dim, hidden_dim), (hidden_dim, embedding_dim)]
        for k, (fi, fo) in enumerate(dims, 1):
            if gcn_name == 'SAGE':
                conv = DenseSAGEConv(fi, fo, normalize=True, bias=bias)
            else:  # model/network.py:97-99: note BOTH Linears map to hidden_dim -> 'fo' below
                conv = DenseGINConv(nn.Sequential(nn.Linear(fi, fo), make_activation(activation),
                                                  nn.Linear(fo, fo)))
            setattr(self, 'gcn%d' % k, conv)
            if bn:
                setattr(self, 'bn%d' % k, nn.BatchNorm1d(fo))
        self.use_bn = bn
        self.act = make_activation(activation)
        self.lin = nn.Linear(2 * hidden_dim + embedding_dim, embedding_dim) if lin else None

    def _bn(self, k, h):
        # model/network.py:101-107: statistics over ALL B*Nmax rows, padded zero rows included.
        if not self.use_bn:
            return h
        b, n, c = h.shape
        return getattr(self, 'bn%d' % k)(h.reshape(-1, c)).reshape(b, n, c)

    def forward(self, x, adj, mask=None):
        outs = []
        h = x
        for k in (1, 2, 3):
            h = self._bn(k, self.act(getattr(self, 'gcn%d' % k)(h, adj, mask, self.add_loop)))
            outs.append(h)
        h = torch.cat(outs, dim=-1)
        if mask is not None:
            h = h * mask
        if self.lin is not None:
            h = self.lin(h)
            if mask is not None:
                h = h * mask
        return h


class DenseJK(nn.Module):
    """model/network.py:11-55: bi-LSTM attention over the three layer outputs of a block."""

    def __init__(self, channels, num_layers=3):
        super().__init__()
        self.channel = channels
        hid = channels * num_layers // 2
        self.lstm = nn.LSTM(channels, hid, bidirectional=True, batch_first=True)
        self.att = nn.Linear(2 * hid, 1)

    def forward(self, xs):
        b, n, _ = xs.shape
        seq = torch.stack(torch.split(xs, self.channel, dim=-1), dim=2).reshape(b * n, -1, self.channel)
        alpha, _ = self.lstm(seq)
        alpha = torch.softmax(self.att(alpha).squeeze(-1), dim=-1)
        return (seq * alpha.unsqueeze(-1)).sum(dim=1).reshape(b, n, self.channel)


class SoftPoolingGcnEncoder(nn.Module):
    """model/network.py:127-291.  Same constructor signature, same state_dict keys, CPU, dense."""

    def __init__(self, max_num_nodes, input_dim, hidden_dim, embedding_dim, bias, bn, assign_hidden_dim,
                 label_dim, assign_ratio=0.25, pred_hidden_dims=(50,), concat=True, gcn_name='SAGE',
                 collect_assign=False, load_data_sparse=False, norm_adj=False, activation='relu',
                 drop_out=0., jk=False):
        super().__init__()
        self.jk, self.drop_out, self.norm_adj = jk, drop_out, norm_adj
        self.load_data_sparse, self.collect_assign = load_data_sparse, collect_assign
        self.assign_matrix = []
        c1 = int(max_num_nodes * assign_ratio)          # model/network.py:142
        c2 = int(c1 * assign_ratio)                     # model/network.py:155
        kw = dict(bias=bias, bn=bn, add_loop=False, gcn_name=gcn_name, activation=activation)
        self.GCN_embed_1 = GNNBlock(input_dim, hidden_dim, embedding_dim, lin=False, **kw)
        if jk:
            self.jk1 = DenseJK(hidden_dim, 3)
        self.GCN_pool_1 = GNNBlock(input_dim, assign_hidden_dim, c1, lin=True, **kw)
        dx = hidden_dim * 2 + embedding_dim if (concat and not jk) else embedding_dim  # :150-153
        self.GCN_embed_2 = GNNBlock(dx, hidden_dim, embedding_dim, lin=False, **kw)
        if jk:
            self.jk2 = DenseJK(hidden_dim, 3)
        self.GCN_pool_2 = GNNBlock(dx, assign_hidden_dim, c2, lin=True, **kw)
        self.GCN_embed_3 = GNNBlock(dx, hidden_dim, embedding_dim, lin=False, **kw)
        if jk:
            self.jk3 = DenseJK(hidden_dim, 3)
        # model/network.py:220-234
        layers, d = [], dx * 3
        pred_hidden_dims = list(pred_hidden_dims)
        if not pred_hidden_dims:
            self.pred_model = nn.Linear(d, label_dim)
        else:
            for h in pred_hidden_dims:
                layers += [nn.Linear(d, h), make_activation(activation)]
                d = h
                if drop_out > 0:
                    layers.append(nn.Dropout(drop_out))
            layers.append(nn.Linear(d, label_dim))
            self.pred_model = nn.Sequential(*layers)

    def _stage(self, k, x, adj, mask):
        embed = getattr(self, 'GCN_embed_%d' % k)(x, adj, mask)
        if self.jk:
            embed = getattr(self, 'jk%d' % k)(embed)
        return embed, embed.max(dim=1)[0]   # max readout over ALL rows incl. zero padding (:264)

    def forward(self, data):
        self.assign_matrix = []
        if self.load_data_sparse:           # model/network.py:237-243
            adj = to_dense_adj(data.edge_index, data.batch)
            x, counts = to_dense_batch(data.x, data.batch)
            label = data.y
        else:                               # model/network.py:253-256
            x, adj, counts = data[0], data[1], data[2]
            label = data[3] if self.training else None
        mask = node_mask(adj.shape[1], counts)
        readouts = []
        for level in (1, 2, 3):
            if self.norm_adj:
                adj = re_norm_adj(adj, 0.4, mask)
            embed, ro = self._stage(level, x, adj, mask)
            readouts.append(ro)
            if level < 3:
                assign = getattr(self, 'GCN_pool_%d' % level)(x, adj, mask)
                x, adj, s_soft = diff_pool(embed, adj, assign, mask)
                if self.collect_assign:
                    self.assign_matrix.append(s_soft.detach())
                mask = None
        logits = self.pred_model(torch.cat(readouts, dim=1))
        if self.training:
            return logits, F.cross_entropy(logits, label)  # size_average=True == mean (:289)
        return logits
