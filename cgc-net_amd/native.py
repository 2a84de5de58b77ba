"""Python face of the step sequencer (csrc/exec.hip; include/cgc_hip.h: cgc_level_fwd / cgc_level_bwd).

One ``torch.autograd.Function`` per LEVEL of ``SoftPoolingGcnEncoder.forward`` (model/network.py:258-268, :270-278, :279-285)
instead of one per operator: Python issues 3 + 3 library calls per training step where the per-operator path (ops.py) issues
~330, the schedule inside a level is C++.  The arithmetic is the per-operator path's, kernel for kernel (tests compare the two
bit for bit); what autograd sees is (readout, pooled features, pooled adjacency) per level and one flat gradient buffer whose
slices are the parameters' gradients.

Scope (``supported``): DenseSAGEConv blocks in training mode (or without BatchNorm), fixed BatchNorm momentum, DenseJK on the
compiled channel counts, input features without gradient.  Everything else runs on the per-operator path.
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import kernels
from .graph import uniform_ptr
from .kernels import ACT_CODES

P, I, F_, D_, L = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64


class LevelDesc(C.Structure):
    _fields_ = [('level', I), ('B', I), ('n', I), ('rows_per_graph', I), ('nmax', I), ('npad', I), ('fin', I), ('H', I), ('E', I),
                ('AH', I), ('C', I), ('has_bias', I), ('has_bn', I), ('act', I), ('jk', I), ('renorm', I), ('renorm_p', F_),
                ('bn_eps', F_ * 6), ('bn_momentum', F_ * 6), ('count', D_), ('eval', I), ('flags', I)]


class BlockParams(C.Structure):
    _fields_ = [('W', P * 3), ('b', P * 3), ('gamma', P * 3), ('beta', P * 3), ('running_mean', P * 3), ('running_var', P * 3),
                ('num_batches_tracked', P * 3), ('lin_W', P), ('lin_b', P)]


class JkParams(C.Structure):
    _fields_ = [('lstm', P * 8), ('w_att', P), ('b_att', P)]


class Graph(C.Structure):
    _fields_ = [('rowptr', P), ('col', P), ('t_rowptr', P), ('t_col', P), ('val', P), ('t_val', P), ('inv_d', P), ('gorder', P), ('spatial', I)]


class GradLayout(C.Structure):
    _fields_ = [('W', L * 6), ('b', L * 6), ('bn_weight', L * 6), ('bn_bias', L * 6), ('lin_W', L), ('lin_b', L), ('jk', L),
                ('total', L)]


def _lib():
    return kernels.get().lib       # prototypes: _abi.py


def _p(t):
    return t.data_ptr() if t is not None else None


# ---- which tensors of a module go where (fixed order: the Function's inputs, the gradient slices) --------------------------
def _block_tensors(blk):
    """[W1, b1, W2, b2, W3, b3, bn1.weight, bn1.bias, .., bn3.bias, lin.weight, lin.bias] (None where the module has none)."""
    out = []
    for k in (1, 2, 3):
        conv = getattr(blk, 'gcn%d' % k)
        out += [conv.weight, conv.bias]
    for k in (1, 2, 3):
        bn = getattr(blk, 'bn%d' % k) if blk.use_bn else None
        out += [bn.weight if bn is not None else None, bn.bias if bn is not None else None]
    lin = blk.lin
    out += [lin.weight if lin is not None else None, lin.bias if lin is not None else None]
    return out


def _jk_tensors(jk):
    p = jk.lstm
    return [p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0, p.weight_ih_l0_reverse, p.weight_hh_l0_reverse,
            p.bias_ih_l0_reverse, p.bias_hh_l0_reverse, jk.att.weight, jk.att.bias]


def _block_params(blk):
    s = BlockParams()
    if blk is None:
        return s
    for k in range(3):
        conv = getattr(blk, 'gcn%d' % (k + 1))
        s.W[k], s.b[k] = _p(conv.weight), _p(conv.bias)
        if blk.use_bn:
            bn = getattr(blk, 'bn%d' % (k + 1))
            s.gamma[k], s.beta[k] = _p(bn.weight), _p(bn.bias)
            track = bn.track_running_stats and bn.running_mean is not None
            s.running_mean[k] = _p(bn.running_mean) if track else None
            s.running_var[k] = _p(bn.running_var) if track else None
            s.num_batches_tracked[k] = _p(bn.num_batches_tracked) if track else None
    if blk.lin is not None:
        s.lin_W, s.lin_b = _p(blk.lin.weight), _p(blk.lin.bias)
    return s


def _jk_params(jk):
    s = JkParams()
    if jk is None:
        return s
    t = _jk_tensors(jk)
    for i in range(8):
        s.lstm[i] = _p(t[i])
    s.w_att, s.b_att = _p(t[8]), _p(t[9])
    return s


def _f32ok(*ts):
    return all(t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda) for t in ts)


def _mode_modules(blk):
    """The modules of a block whose .training flag decides what the sequencer computes (BatchNorm on batch or running statistics)."""
    return [blk] + ([getattr(blk, 'bn%d' % k) for k in (1, 2, 3)] if blk.use_bn else [])


def prepared(enc, level, emb, pool, jk, fin):
    """Everything about a level that does not change from batch to batch: whether the sequencer covers it, a LevelDesc template,
    the parameter tensors in the Function's order and the pointer structs -- ~0.4 ms of Python per step when it was redone for
    every call.  Cached on the encoder and keyed by what would invalidate it: the parameter objects and their storage, the mode
    flags (train / eval, norm_adj), BatchNorm's eps / momentum."""
    blocks = [emb] + ([pool] if pool is not None else [])
    if any(not b.mean_aggregation or b.add_loop for b in blocks) or (jk is not None and jk.mode != 'lstm'):
        return None                                   # (GIN blocks have no .weight to list)
    params = _block_tensors(emb) + (_block_tensors(pool) if pool is not None else []) + (_jk_tensors(jk) if jk is not None else [])
    bn_cfg = tuple((getattr(b, 'bn%d' % k).eps, getattr(b, 'bn%d' % k).momentum, getattr(b, 'bn%d' % k).track_running_stats)
                   for b in blocks if b.use_bn for k in (1, 2, 3))
    modes = (enc.training,) + tuple(m.training for b in blocks for m in _mode_modules(b))
    key = (tuple(id(t) for t in params), tuple(t.data_ptr() for t in params if t is not None), modes, enc.norm_adj, fin, bn_cfg,
           emb.activation, int(getattr(enc, 'gemm_mode', 0)))
    cache = enc.__dict__.setdefault('_native_prepared', {})
    hit = cache.get(level)
    if hit is not None and hit[0] == key:
        return hit[1]
    d = describe(enc, level, emb, pool, jk, 1, 1, 1 if level >= 2 else 0, 1, 1, fin, 1.0, check_lib=False)
    res = None
    if d is not None:
        res = dict(template=bytes(d), params=params, structs=(_block_params(emb), _block_params(pool), _jk_params(jk)))
    cache[level] = (key, res)
    return res


def describe_from(prep, level, B, n, rows_per_graph, nmax, npad, count):
    """The batch's LevelDesc from the cached template, or None when the library refuses the dimensions."""
    d = LevelDesc.from_buffer_copy(prep['template'])
    d.level, d.B, d.n, d.rows_per_graph, d.nmax, d.npad, d.count = level, B, n, rows_per_graph, nmax, npad, float(count)
    return d if _lib().cgc_level_supported(C.byref(d)) else None


def describe(enc, level, emb, pool, jk, B, n, rows_per_graph, nmax, npad, fin, count, check_lib=True):
    """LevelDesc for one level, or None when the sequencer does not cover the configuration (the caller then takes the
    per-operator path)."""
    blocks = [emb] + ([pool] if pool is not None else [])
    # ONE train / eval decision per level, taken from the encoder (network._native_level dispatches on enc.training): a block or a
    # BatchNorm in the other mode (a frozen block while fine-tuning) is the per-operator path's business -- the sequencer would size
    # its scratch for one mode and normalise with the other's statistics
    if any(m.training != enc.training for blk in blocks for m in _mode_modules(blk)):
        return None
    for blk in blocks:
        if not blk.mean_aggregation or blk.add_loop:
            return None
        for k in (1, 2, 3):
            conv = getattr(blk, 'gcn%d' % k)
            if not conv.normalize or (conv.bias is None) != (emb.gcn1.bias is None):
                return None
            if blk.use_bn:
                bn = getattr(blk, 'bn%d' % k)
                if not bn.affine:
                    return None
                if blk.training and bn.momentum is None:
                    return None
                if not blk.training and (not bn.track_running_stats or bn.running_mean is None):
                    return None            # (inference without running statistics normalises with batch statistics: per-operator path)
        if blk.use_bn != emb.use_bn or blk.activation != emb.activation:
            return None
        if not _f32ok(*_block_tensors(blk)):
            return None
    if jk is not None and (jk.mode != 'lstm' or not _f32ok(*_jk_tensors(jk))):
        return None
    if pool is not None and (pool.lin is None or pool.gcn1.out_channels != pool.gcn2.out_channels):
        return None
    if emb.lin is not None or emb.gcn1.out_channels != emb.gcn2.out_channels:
        return None
    d = LevelDesc()
    d.level, d.B, d.n, d.rows_per_graph, d.nmax, d.npad, d.fin = level, B, n, rows_per_graph, nmax, npad, fin
    d.H, d.E = emb.gcn1.out_channels, emb.gcn3.out_channels
    d.AH, d.C = (pool.gcn1.out_channels, pool.gcn3.out_channels) if pool is not None else (0, 0)
    d.has_bias, d.has_bn = int(emb.gcn1.bias is not None), int(emb.use_bn)
    d.act, d.jk = ACT_CODES[emb.activation], int(jk is not None)
    d.renorm, d.renorm_p = int(enc.norm_adj), float(RENORM_P)
    d.eval = int(not enc.training)
    d.flags = {0: 0, 1: 2, 2: 4}[int(getattr(enc, 'gemm_mode', 0))]      # bit 1: six bf16 pairs, bit 2: three fp16 pairs (bit 0 is reserved: include/cgc_hip.h)
    for b_i, blk in enumerate(blocks):
        if blk.use_bn:
            for k in range(3):
                bn = getattr(blk, 'bn%d' % (k + 1))
                # (momentum None = cumulative average: refused above in training; unused in inference)
                d.bn_eps[3 * b_i + k], d.bn_momentum[3 * b_i + k] = bn.eps, 0.0 if bn.momentum is None else bn.momentum
    d.count = float(count)
    if emb.gcn1.in_channels != fin or (pool is not None and pool.gcn1.in_channels != fin):
        return None
    if check_lib and not _lib().cgc_level_supported(C.byref(d)):
        return None
    return d


RENORM_P = 0.4      # model/network.py:260,271,280


_SIZES = {}


def _sizes(d):
    """(saved floats, scratch floats, gradient layout) of a LevelDesc; memoised on the descriptor's bytes (bounded: the level-1
    descriptor changes with every batch's node count)."""
    key = bytes(d)
    hit = _SIZES.get(key)
    if hit is None:
        lib = _lib()
        lay = GradLayout()
        lib.cgc_level_grad_layout_of(C.byref(d), C.byref(lay))
        hit = (int(lib.cgc_level_saved_floats(C.byref(d))), int(lib.cgc_level_scratch_floats(C.byref(d))), lay)
        if len(_SIZES) > 256:
            _SIZES.clear()
        _SIZES[key] = hit
    return hit


class _Level(Function):
    """(readout, x_out, A_out) = one level; ``cfg`` carries the non-tensor arguments."""

    @staticmethod
    def forward(ctx, cfg, x_in, A_in, *params):
        lib = _lib()
        d, emb, pool, jk, g, gptr = cfg['desc'], cfg['emb'], cfg['pool'], cfg['jk'], cfg['graph'], cfg['gptr']
        dev = x_in.device
        kernels.get()._dev(x_in, A_in, gptr)
        stream = kernels.get()._stream()
        n_saved, n_scratch, _ = _sizes(d)
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        D = d.H if d.jk else 2 * d.H + d.E
        readout = torch.empty(d.B, D, dtype=torch.float32, device=dev)
        x_out = torch.empty(d.B, d.C, D, dtype=torch.float32, device=dev) if d.C else None
        A_out = torch.empty(d.B, d.C, d.C, dtype=torch.float32, device=dev) if d.C else None
        pe, pp, pj = cfg['structs'] if cfg.get('structs') is not None else (_block_params(emb), _block_params(pool), _jk_params(jk))
        gs = Graph()
        if g is not None:
            gs.rowptr, gs.col, gs.t_rowptr, gs.t_col = _p(g.rowptr), _p(g.col), _p(g.t_rowptr), _p(g.t_col)
            gs.val, gs.t_val, gs.inv_d, gs.gorder = _p(g.val), _p(g.t_val), _p(g.inv_d), _p(g.gorder)
            gs.spatial = int(bool(getattr(g, 'spatial', False)))
        s_ptr, s_ld = P(), I()
        rc = lib.cgc_level_fwd(C.byref(d), C.byref(pe), C.byref(pp), C.byref(pj), C.byref(gs), _p(gptr), _p(x_in), _p(A_in), _p(saved),
                               _p(scratch), _p(readout), _p(x_out), _p(A_out), C.byref(s_ptr), C.byref(s_ld), stream)
        if rc != 0:
            raise RuntimeError('cgc_level_fwd failed with code %d' % rc)
        if cfg.get('assign') is not None and d.C:
            off = (s_ptr.value - saved.data_ptr()) // 4
            cfg['assign'].append(torch.as_strided(saved, (d.n, d.C), (s_ld.value, 1), off).detach().clone())
        ctx.cfg, ctx.structs = cfg, (pe, pp, pj, gs)
        ctx.save_for_backward(x_in, A_in, saved, *[p for p in params if p is not None])
        ctx.mask = [p is not None for p in params]
        ctx.shapes = [tuple(p.shape) if p is not None else None for p in params]
        if d.C:
            return readout, x_out, A_out
        return readout

    @staticmethod
    def backward(ctx, d_readout, d_x_out=None, d_A_out=None):
        lib = _lib()
        cfg = ctx.cfg
        d, g, gptr = cfg['desc'], cfg['graph'], cfg['gptr']
        x_in, A_in, saved = ctx.saved_tensors[:3]
        pe, pp, pj, gs = ctx.structs
        dev = saved.device
        stream = kernels.get()._stream()
        _, n_scratch, lay = _sizes(d)
        grads = _grad_buffer(cfg.get('owner'), d.level, int(lay.total), dev)
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        D = d.H if d.jk else 2 * d.H + d.E
        d_readout = d_readout.contiguous().float()
        if d.C:
            d_x_out = torch.zeros(d.B, d.C, D, device=dev) if d_x_out is None else d_x_out.contiguous().float()
            d_A_out = torch.zeros(d.B, d.C, d.C, device=dev) if d_A_out is None else d_A_out.contiguous().float()
        dense = d.level >= 2
        d_x_in = torch.empty_like(x_in) if dense else None
        d_A_in = torch.empty_like(A_in) if dense else None
        rc = lib.cgc_level_bwd(C.byref(d), C.byref(pe), C.byref(pp), C.byref(pj), C.byref(gs), _p(gptr), _p(x_in), _p(A_in), _p(saved),
                               _p(scratch), _p(d_readout), _p(d_x_out), _p(d_A_out), _p(grads), _p(d_x_in), _p(d_A_in), stream)
        if rc != 0:
            raise RuntimeError('cgc_level_bwd failed with code %d' % rc)
        owner = cfg.get('owner')
        if owner is not None:            # the flat buffer of this level's gradients, for the one-launch optimiser (optim.Adam)
            owner._flat_grads[d.level] = grads
        out = []
        offs = _param_offsets(d, lay)
        for present, shape, off in zip(ctx.mask, ctx.shapes, offs):
            if not present or off < 0:
                out.append(None)
                continue
            k = 1
            for s_ in shape:
                k *= s_
            out.append(grads[off:off + k].view(shape))
        return (None, d_x_in, d_A_in) + tuple(out)


def _param_offsets(d, lay):
    """Gradient-buffer offsets in the order of the Function's parameter inputs (see level())."""
    offs = []
    nblk = 2 if d.C else 1
    for b_i in range(nblk):
        for k in range(3):
            offs += [lay.W[3 * b_i + k], lay.b[3 * b_i + k]]
        for k in range(3):
            offs += [lay.bn_weight[3 * b_i + k], lay.bn_bias[3 * b_i + k]]
        offs += ([lay.lin_W, lay.lin_b] if b_i == 1 else [-1, -1])
    if d.jk:
        H = d.H
        Hh = 3 * H // 2
        o = lay.jk
        for _ in range(2):
            for size in (4 * Hh * H, 4 * Hh * Hh, 4 * Hh, 4 * Hh):
                offs.append(o)
                o += size
        offs += [o, o + 2 * Hh]
    return offs


def level(enc, desc, emb, pool, jk, g, gptr, x_in, A_in, assign=None, prep=None):
    """Run one level through the sequencer.  Returns (readout, x_out, A_out) (x_out / A_out None at the last level)."""
    cfg = dict(desc=desc, emb=emb, pool=pool, jk=jk, graph=g, gptr=gptr, assign=assign, structs=prep['structs'] if prep else None,
               owner=enc)
    params = prep['params'] if prep else (_block_tensors(emb) + (_block_tensors(pool) if pool is not None else []) +
                                          (_jk_tensors(jk) if jk is not None else []))
    _register_flat(enc, desc.level, params, lambda: _param_offsets(desc, _sizes(desc)[2]), lambda: _sizes(desc)[2].total)
    out = _Level.apply(cfg, x_in, A_in, *params)
    if desc.C:
        return out
    return out, None, None


def level_eval(enc, desc, emb, pool, jk, g, gptr, x_in, A_in, assign=None, prep=None):
    """Inference forward of one level through the sequencer (desc.eval = 1: BatchNorm on its running statistics, nothing kept): no
    autograd node, two arenas that die with the call.  Returns (readout, x_out, A_out).  evaluate() (train.py:21-91) runs on this."""
    lib = _lib()
    K = kernels.get()
    d = desc
    dev = x_in.device
    K._dev(x_in, A_in, gptr)
    n_saved, n_scratch, _ = _sizes(d)
    saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
    scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
    D = d.H if d.jk else 2 * d.H + d.E
    readout = torch.empty(d.B, D, dtype=torch.float32, device=dev)
    x_out = torch.empty(d.B, d.C, D, dtype=torch.float32, device=dev) if d.C else None
    A_out = torch.empty(d.B, d.C, d.C, dtype=torch.float32, device=dev) if d.C else None
    pe, pp, pj = prep['structs'] if prep else (_block_params(emb), _block_params(pool), _jk_params(jk))
    gs = Graph()
    if g is not None:
        gs.rowptr, gs.col, gs.t_rowptr, gs.t_col = _p(g.rowptr), _p(g.col), _p(g.t_rowptr), _p(g.t_col)
        gs.val, gs.t_val, gs.inv_d, gs.gorder = _p(g.val), _p(g.t_val), _p(g.inv_d), _p(g.gorder)
        gs.spatial = int(bool(getattr(g, 'spatial', False)))
    s_ptr, s_ld = P(), I()
    rc = lib.cgc_level_fwd(C.byref(d), C.byref(pe), C.byref(pp), C.byref(pj), C.byref(gs), _p(gptr), _p(x_in), _p(A_in), _p(saved),
                           _p(scratch), _p(readout), _p(x_out), _p(A_out), C.byref(s_ptr), C.byref(s_ld), K._stream())
    if rc != 0:
        raise RuntimeError('cgc_level_fwd (inference) failed with code %d' % rc)
    if assign is not None and d.C:
        off = (s_ptr.value - saved.data_ptr()) // 4
        assign.append(torch.as_strided(saved, (d.n, d.C), (s_ld.value, 1), off).clone())
    return readout, x_out, A_out


def _grad_buffer(owner, slot, n, dev):
    """The flat gradient buffer of one level / of the head.  With an owner whose four sizes are known, the four are slices of ONE
    buffer per backward pass (head first, level 1 last; each slice starts on a 256-byte boundary): data parallelism all-reduces that
    one buffer in place and the optimiser reads it where it is.  A slot asked for twice means a new backward pass has begun."""
    sizes = getattr(owner, '_flat_sizes', None) if owner is not None else None
    if not sizes or len(sizes) != 4 or sizes.get(slot) != n:
        return torch.empty(n, dtype=torch.float32, device=dev)
    st = owner.__dict__.get('_step_flat')
    if st is None or slot in owner._step_taken or st.device != dev:
        total = sum(-(-sizes[s_] // 64) * 64 for s_ in range(4))
        st = owner._step_flat = torch.empty(total, dtype=torch.float32, device=dev)
        owner._step_taken = set()
    owner._step_taken.add(slot)
    o = sum(-(-sizes[s_] // 64) * 64 for s_ in range(slot))
    return st[o:o + n]


def _register_flat(enc, slot, params, offsets, total=None):
    """Where the gradient of every parameter of this level (slot 1..3) / of the head (slot 0) will be found: (slot, element offset
    into the slot's flat buffer).  Static for the life of the model -- the layout depends on widths only -- so it is recorded once."""
    if enc is None:
        return
    if not hasattr(enc, '_flat_index'):
        enc._flat_index, enc._flat_grads, enc._flat_sizes = {}, {}, {}
    if slot in enc._flat_index:
        return
    enc._flat_index[slot] = [(p, int(off)) for p, off in zip(params, offsets()) if p is not None and off >= 0]
    enc._flat_sizes[slot] = int(total())


def static_flat(enc):
    """Register the flat gradient layout of a model WITHOUT running it (it depends on widths only) and return the number of floats
    of the per-pass gradient buffer (native._grad_buffer), or None when the step sequencer does not cover the model as configured.
    parallel.DataParallel sizes its all-reduce by it, so that every rank issues the same collective whether its gradients came
    out of the sequencer this step or not (a rank that idles through a step; a pass on the per-operator path)."""
    import torch.nn as nn
    if not getattr(enc, 'native', False) or not getattr(enc, 'native_head', False) or not enc.training:
        return None
    for level in (1, 2, 3):
        emb = getattr(enc, 'GCN_embed_%d' % level, None)
        pool = getattr(enc, 'GCN_pool_%d' % level, None) if level < 3 else None
        jk = getattr(enc, 'jk%d' % level, None) if getattr(enc, 'jk', False) else None
        if emb is None or not hasattr(emb, 'gcn1') or not hasattr(emb.gcn1, 'in_channels'):
            return None
        prep = prepared(enc, level, emb, pool, jk, emb.gcn1.in_channels)
        if prep is None:
            return None
        d = LevelDesc.from_buffer_copy(prep['template'])
        lay = _sizes(d)[2]
        _register_flat(enc, level, prep['params'], lambda d=d, lay=lay: _param_offsets(d, lay), lambda lay=lay: lay.total)
    layers = list(enc.pred_model) if isinstance(enc.pred_model, nn.Sequential) else [enc.pred_model]
    if len(layers) not in (3, 4) or not isinstance(layers[0], nn.Linear) or not isinstance(layers[-1], nn.Linear):
        return None
    l1, l2 = layers[0], layers[-1]
    H1, Kin, L_ = l1.out_features, l1.in_features, l2.out_features
    _register_flat(enc, 0, [l1.weight, l1.bias, l2.weight, l2.bias], lambda: [0, H1 * Kin, H1 * Kin + H1, H1 * Kin + H1 + L_ * H1],
                   lambda: H1 * Kin + H1 + L_ * H1 + L_)
    if len(enc._flat_sizes) != 4:
        return None
    return sum(-(-enc._flat_sizes[s_] // 64) * 64 for s_ in range(4))


def flat_slot_offset(enc, slot):
    """First float of a slot inside the per-pass gradient buffer."""
    return sum(-(-enc._flat_sizes[s_] // 64) * 64 for s_ in range(slot))


def dense_gptr(B, Cn, device):
    return uniform_ptr(B, Cn, device)


# ---- classification head + loss as one kernel each way (csrc/head.hip) ---------------------------------------------------------
class _Head(Function):
    """(logits, loss) = head(readouts); ``cfg`` = dict(act, drop_p, seed, labels)."""

    @staticmethod
    def forward(ctx, cfg, W1, b1, W2, b2, *xs):
        lib = _lib()
        K = kernels.get()
        K._dev(W1, b1, W2, b2, cfg['labels'], *xs)
        xs = [x.contiguous() for x in xs]
        B, D = xs[0].shape
        H1, L_ = W1.shape[0], W2.shape[0]
        dev = xs[0].device
        ws = torch.empty(3 * B * H1 + B, dtype=torch.float32, device=dev)
        logits = torch.empty(B, L_, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        y = cfg['labels']
        px = (P * len(xs))(*[x.data_ptr() for x in xs])
        rc = lib.cgc_head_fwd(px, len(xs), B, D, H1, L_, cfg['act'], _p(W1), _p(b1), _p(W2), _p(b2), _p(y), C.c_float(cfg['drop_p']),
                              C.c_uint64(cfg['seed']), _p(ws), _p(logits), _p(loss), K._stream())
        if rc != 0:
            raise RuntimeError('cgc_head_fwd failed with code %d' % rc)
        ctx.cfg, ctx.dims = cfg, (B, D, H1, L_, len(xs))
        if cfg.get('owner') is not None:       # the workspace of the last head call (z | keep | h | per-sample losses): last_dropout_mask()
            cfg['owner'].__dict__['_last_head'] = (ws, B, H1)
        ctx.save_for_backward(W1, W2, ws, logits, *xs)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.mark_non_differentiable()
        return logits, loss

    @staticmethod
    def backward(ctx, d_logits, d_loss):
        lib = _lib()
        K = kernels.get()
        cfg = ctx.cfg
        B, D, H1, L_, nseg = ctx.dims
        W1, W2, ws, logits = ctx.saved_tensors[:4]
        xs = ctx.saved_tensors[4:]
        dev = ws.device
        Kin = nseg * D
        grads = _grad_buffer(cfg.get('owner'), 0, H1 * Kin + H1 + L_ * H1 + L_, dev)
        scratch = torch.empty(B * L_ + B * H1, dtype=torch.float32, device=dev)
        dxs = [torch.empty(B, D, dtype=torch.float32, device=dev) for _ in range(nseg)]
        d_loss = d_loss.contiguous().float() if d_loss is not None else None
        d_logits = d_logits.contiguous().float() if d_logits is not None else None
        px = (P * nseg)(*[x.data_ptr() for x in xs])
        pdx = (P * nseg)(*[x.data_ptr() for x in dxs])
        rc = lib.cgc_head_bwd(px, nseg, B, D, H1, L_, cfg['act'], _p(W1), _p(W2), _p(cfg['labels']), _p(ws), _p(logits), _p(d_loss),
                              _p(d_logits), _p(scratch), _p(grads), pdx, K._stream())
        if rc != 0:
            raise RuntimeError('cgc_head_bwd failed with code %d' % rc)
        owner = cfg.get('owner')
        if owner is not None:
            owner._flat_grads[0] = grads
        o = 0
        dW1 = grads[o:o + H1 * Kin].view(H1, Kin)
        o += H1 * Kin
        db1 = grads[o:o + H1] if ctx.has_bias[0] else None
        o += H1
        dW2 = grads[o:o + L_ * H1].view(L_, H1)
        o += L_ * H1
        db2 = grads[o:o + L_] if ctx.has_bias[1] else None
        return (None, dW1, db1, dW2, db2) + tuple(dxs)


def last_dropout_mask(enc):
    """The dropout keep-scale plane [B, H1] the fused head applied in ``enc``'s most recent training forward: 0 where a hidden unit
    was dropped, 1 / (1 - p) where it was kept (all ones without dropout).  A view of the head's saved workspace (csrc/head.hip:
    ``keep``): the mask-exact parity tests apply THIS mask to the module stack and to the oracle."""
    ws, B, H1 = enc.__dict__['_last_head']
    return ws[B * H1:2 * B * H1].view(B, H1)


# per-encoder caches that must not travel with copy.deepcopy / pickle (ctypes structs hold raw pointers; the rest is rebuilt on use)
TRANSIENT = ('_native_prepared', '_flat_index', '_flat_grads', '_flat_sizes', '_step_flat', '_step_taken', '_last_head')


def head(pred_model, readouts, labels, training, owner=None):
    """``pred_model(cat(readouts))`` + mean cross-entropy through the fused head kernels; None when the head is not the
    Linear -> activation -> [Dropout] -> Linear stack they cover (the caller then runs the modules one by one)."""
    import torch.nn as nn
    layers = list(pred_model) if isinstance(pred_model, nn.Sequential) else [pred_model]
    if len(layers) not in (3, 4) or not isinstance(layers[0], nn.Linear) or not isinstance(layers[-1], nn.Linear):
        return None
    act = {nn.ReLU: 'relu', nn.ELU: 'elu', nn.LeakyReLU: 'leakyrelu'}.get(type(layers[1]))
    if act is None or (act == 'elu' and layers[1].alpha != 1.0) or (act == 'leakyrelu' and layers[1].negative_slope != 0.01):
        return None
    drop_p = 0.0
    if len(layers) == 4:
        if not isinstance(layers[2], nn.Dropout):
            return None
        drop_p = float(layers[2].p) if training else 0.0
    if len(readouts) > 3 or any(r.shape != readouts[0].shape for r in readouts) or not (0.0 <= drop_p < 1.0):
        return None
    l1, l2 = layers[0], layers[-1]
    if l1.in_features != len(readouts) * readouts[0].shape[1] or not _f32ok(l1.weight, l1.bias, l2.weight, l2.bias):
        return None
    # the mask is a function of (seed, element): the seed comes from torch's CPU generator (no device work; torch.manual_seed governs it)
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) if drop_p > 0.0 else 0
    cfg = dict(act=ACT_CODES[act], drop_p=drop_p, seed=seed, labels=labels.view(-1).contiguous(), owner=owner)
    H1, Kin, L_ = l1.out_features, l1.in_features, l2.out_features
    _register_flat(owner, 0, [l1.weight, l1.bias, l2.weight, l2.bias], lambda: [0, H1 * Kin, H1 * Kin + H1, H1 * Kin + H1 + L_ * H1],
                   lambda: H1 * Kin + H1 + L_ * H1 + L_)
    return _Head.apply(cfg, l1.weight, l1.bias, l2.weight, l2.bias, *readouts)
