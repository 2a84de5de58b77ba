"""The reference's optimiser (common/utils.py:119-121: Adam, lr 1e-3, weight decay 1e-4) without torch's per-step bookkeeping.

``torch.optim.Adam(fused=True).step()`` regroups ~100 parameters by device and dtype and rebuilds five lists on every call
(~0.25 ms of host time; at 4 graphs per GPU the GPU waits for it between the end of backward and the update).  Two levels:

* the lists do not change from step to step, so they are built once; ``step`` then is ``torch._foreach_add_(steps, 1)`` +
  ``torch._fused_adam_`` -- the same kernels with the same arguments torch's own ``step`` ends in (any model);
* ``Adam(params, model=encoder)``: when the step sequencer produced the gradients (network.SoftPoolingGcnEncoder on its default
  path), every parameter's gradient sits at a fixed offset of one of four flat buffers (native._register_flat), and the whole
  update is ONE launch of the library's ``cgc_adam_step`` over a segment table built once: no lists, no step-counter kernel, no
  per-tensor metadata.  The arithmetic is torch's fused kernel's, bit for bit (tests/test_native_gpu.py).  Anything unexpected --
  a gradient that is not where the sequencer leaves it (accumulation over several backward passes, a parameter trained through
  the per-operator path), parameters moved, AMSGrad, ... -- falls back to the level above for that step.

State, ``state_dict`` and LR schedulers are torch's (the per-parameter ``step`` tensors are brought up to date before anything reads
them).
"""
import ctypes as C

import torch


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, model=None, grad_mul=1.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self._lists = None
        self._model = model
        self._table = None          # (segs, blocks, nblocks, sentinels) of the one-launch path
        self._ptrs = None           # parameter addresses the table was built for
        self._t = None              # step count of the one-launch path (None: torch's step tensors are current)
        self._uneven = False        # per-parameter step counts differ: the one-launch kernel (one count for all) is not used
        # every gradient is multiplied by grad_mul inside the update (p.grad itself is left as it is).  For callers that keep SUMMED
        # gradients and want the mean taken here; parallel.DataParallel does NOT use it -- it hands over averaged gradients
        self.grad_mul = grad_mul

    # ---- torch's fused kernel on cached lists
    def _cache(self):
        ps = [p for p in self.param_groups[0]['params'] if p.grad is not None]
        st = [self.state[p] for p in ps]
        if not ps or any('exp_avg' not in s for s in st) or len({(p.device, p.dtype) for p in ps}) != 1:
            return None
        return (ps, [s['exp_avg'] for s in st], [s['exp_avg_sq'] for s in st], [s['step'] for s in st],
                len(self.param_groups[0]['params']))

    # ---- the library's one-launch kernel on the sequencer's flat gradient buffers
    def _build_table(self):
        m = self._model
        index = getattr(m, '_flat_index', None)
        if not index or any(s not in (0, 1, 2, 3) for s in index):
            return False                  # (False: this model's gradients do not come out of the sequencer -- do not try again)
        where = {}
        for slot, items in index.items():
            for p, off in items:
                where[id(p)] = (slot, off)
        ps = self.param_groups[0]['params']
        if any(id(p) not in where or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or 'exp_avg' not in self.state[p]
               for p in ps):
            return False
        dev = ps[0].device
        segs, blocks, sentinels, seen = [], [], {}, set()
        for i, p in enumerate(ps):
            slot, off = where[id(p)]
            st = self.state[p]
            segs += [p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), off, p.numel(), slot]
            blocks += [(i, c) for c in range(-(-p.numel() // 1024))]
            if slot not in seen:
                seen.add(slot)
                sentinels[slot] = (p, off, p.data_ptr(), st['exp_avg'].data_ptr())
        self._ptrs = [p.data_ptr() for p in ps]
        seg_t = torch.tensor(segs, dtype=torch.int64).view(-1, 6)
        packed = torch.zeros(len(ps), 6, dtype=torch.int64)        # cgc_adam_seg: 5 x 8 bytes + two int32
        packed[:, :5] = seg_t[:, :5]
        packed[:, 5] = seg_t[:, 5]                                 # slot in the low half (little endian), reserved = 0
        blk = torch.tensor(blocks, dtype=torch.int32).view(-1, 2)
        return (packed.to(dev), blk.to(dev), blk.shape[0], sentinels, dev)

    def _fast_ready(self):
        if not self._table:
            return False
        flat = getattr(self._model, '_flat_grads', None)
        if not flat or self._table[4].index != torch.cuda.current_device():
            return False
        # torch.optim.Adam skips a parameter without a gradient (frozen after the first step: requires_grad = False + zero_grad());
        # the sequencer still writes that parameter's slice of the flat buffer, so the one-launch kernel must not run then
        if any(p.grad is None for p in self.param_groups[0]['params']):
            return False
        # the table holds raw addresses: every parameter must still live where it did (model.to(), p.data = ..., assign=True loads)
        if [p.data_ptr() for p in self.param_groups[0]['params']] != self._ptrs:
            self._table = None               # rebuilt by the next step()
            return False
        for slot, (p, off, pptr, mptr) in self._table[3].items():
            g, f = p.grad, flat.get(slot)
            if g is None or f is None or g.data_ptr() != f.data_ptr() + 4 * off or self.state[p]['exp_avg'].data_ptr() != mptr:
                return False
        return True

    def _scaled_step(self, closure=None):
        """torch's own step with grad_mul applied (the one-launch kernel applies it itself).  The caller's p.grad tensors are left
        untouched: scaled copies stand in for them during the call."""
        if self.grad_mul == 1.0 or closure is not None:
            return super().step(closure)
        ps = [p for gr in self.param_groups for p in gr['params'] if p.grad is not None]
        keep = [p.grad for p in ps]
        for p, g in zip(ps, torch._foreach_mul(keep, self.grad_mul) if keep else []):
            p.grad = g
        try:
            return super().step()
        finally:
            for p, g in zip(ps, keep):
                p.grad = g

    def _flush_steps(self):
        """Bring torch's per-parameter ``step`` tensors up to date with the one-launch path's counter."""
        if self._t is not None:
            steps = [self.state[p]['step'] for p in self.param_groups[0]['params'] if 'step' in self.state[p]]
            if steps:
                torch._foreach_zero_(steps)
                torch._foreach_add_(steps, float(self._t))
            self._t = None

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._t, self._table, self._lists, self._uneven = None, None, None, False
        return super().load_state_dict(state_dict)

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if (closure is not None or len(self.param_groups) != 1 or g.get('amsgrad') or g.get('maximize') or g.get('capturable')
                or g.get('differentiable') or not isinstance(g['lr'], float)):
            self._flush_steps()
            return self._scaled_step(closure)
        # (a parameter that gets its first gradient on a later step, or loses it: the cached lists are rebuilt)
        if self._lists is not None and (self._lists[4] != len(g['params'])
                                        or sum(p.grad is not None for p in g['params']) != len(self._lists[0])):
            self._lists = None
        if self._lists is None:
            self._flush_steps()
            out = self._scaled_step()                 # torch's own path creates the state on the first step
            self._lists = self._cache()
            self._uneven = False                      # re-examined on the next step: uneven step counts can be transient
            if self._model is not None and self._lists is not None and self._table is None:
                self._table = self._build_table()
            return out
        if self._model is not None and self._table is None and self._lists is not None:
            self._table = self._build_table()          # (invalidated: parameters were moved)
        if self._model is not None and not self._uneven and self._t is None and self._fast_ready():
            # (one device read, the first time only)  The kernel takes ONE step count for all parameters: if they differ -- a
            # parameter sat out some steps without a gradient -- the bias corrections differ per parameter and torch's kernel stays
            st = torch.stack([self.state[p]['step'] for p in g['params']])
            lo, hi = float(st.min()), float(st.max())
            if lo != hi:
                self._uneven = True
            else:
                self._t = int(hi)
        if self._model is not None and not self._uneven and self._fast_ready():
            from . import kernels
            self._t += 1
            flat = self._model._flat_grads
            segs, blocks, nblocks, _, dev = self._table
            K = kernels.get()
            bases = (C.c_void_p * 4)(*[flat[s].data_ptr() if s in flat else None for s in range(4)])
            rc = K.lib.cgc_adam_step(C.c_void_p(segs.data_ptr()), C.c_void_p(blocks.data_ptr()), nblocks, bases, g['lr'],
                                     g['betas'][0], g['betas'][1], g['weight_decay'], g['eps'], float(self._t),
                                     float(self.grad_mul), K._stream())
            if rc != 0:
                raise RuntimeError('cgc_adam_step failed with code %d' % rc)
            return None
        self._flush_steps()
        ps, m, v, steps, _ = self._lists
        grads = [p.grad for p in ps]
        if any(x is None for x in grads):
            self._lists = None
            return self._scaled_step()
        if self.grad_mul != 1.0:
            grads = list(torch._foreach_mul(grads, self.grad_mul))
        torch._foreach_add_(steps, 1)
        torch._fused_adam_(ps, grads, m, v, [], steps, amsgrad=False, lr=g['lr'], beta1=g['betas'][0], beta2=g['betas'][1],
                           weight_decay=g['weight_decay'], eps=g['eps'], maximize=False, grad_scale=None, found_inf=None)
        return None
