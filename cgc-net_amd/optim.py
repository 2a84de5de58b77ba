"""The reference's optimiser (common/utils.py:119-121: Adam, lr 1e-3, weight decay 1e-4) as torch's fused kernel without torch's
per-step Python bookkeeping.

``torch.optim.Adam(fused=True).step()`` regroups ~100 parameters by device and dtype and rebuilds five lists on every call
(~0.25 ms of host time; at 4 graphs per GPU the GPU waits for it between the end of backward and the update).  The lists do
not change from step to step, so they are built once; ``step`` then is ``torch._foreach_add_(steps, 1)`` + ``torch._fused_adam_``
-- the same kernels with the same arguments torch's own ``step`` ends in.  State, ``state_dict`` and LR schedulers are torch's.
"""
import torch


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self._lists = None

    def _cache(self):
        ps = [p for p in self.param_groups[0]['params'] if p.grad is not None]
        st = [self.state[p] for p in ps]
        if not ps or any('exp_avg' not in s for s in st) or len({(p.device, p.dtype) for p in ps}) != 1:
            return None
        return (ps, [s['exp_avg'] for s in st], [s['exp_avg_sq'] for s in st], [s['step'] for s in st],
                len(self.param_groups[0]['params']))

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if (closure is not None or len(self.param_groups) != 1 or g.get('amsgrad') or g.get('maximize') or g.get('capturable')
                or g.get('differentiable') or not isinstance(g['lr'], float)):
            return super().step(closure)
        if self._lists is None or self._lists[4] != len(g['params']):
            out = super().step()                      # torch's own path creates the state on the first step
            self._lists = self._cache()
            return out
        ps, m, v, steps, _ = self._lists
        grads = [p.grad for p in ps]
        if any(x is None for x in grads):
            self._lists = None
            return super().step()
        torch._foreach_add_(steps, 1)
        torch._fused_adam_(ps, grads, m, v, [], steps, amsgrad=False, lr=g['lr'], beta1=g['betas'][0], beta2=g['betas'][1],
                           weight_decay=g['weight_decay'], eps=g['eps'], maximize=False, grad_scale=None, found_inf=None)
        return None
