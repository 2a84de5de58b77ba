"""Data parallelism over the GPUs of one node: one process per GPU, RCCL all-reduce of gradients.

Replaces ``torch_geometric.nn.DataParallel`` as the reference uses it (train.py:276-287):
the caller still writes ``model = DataParallel(model); out, loss = model(list_of_Data);
loss.mean().backward(); optimizer.step()`` and ``model.module.state_dict()``.

What the reference does per step inside ONE process (scatter the list over GPUs by cumulative node count,
broadcast all parameters, run replicas on threads, gather, reduce_add gradients onto GPU 0) becomes:
  * every rank holds a replica whose parameters were broadcast ONCE at construction,
  * the list of graphs is split with the same cumulative-node-count rule (data.partition_by_nodes) and each
    rank collates only its chunk (or, with ``shard_input=False``, the loader already hands each rank its own list),
  * graphs are independent, so the forward/backward data path needs no collective,
  * at the end of backward ONE flat fp32 bucket holding every gradient is all-reduced over RCCL/xGMI into the equal-weight
    mean over the replicas that ran -- what the reference takes with ``torch.mean(cls_loss)`` (train.py:179).  On RCCL that
    is ONE collective (``ReduceOp.AVG``: the division happens inside the reduction kernel); a sum followed by a division
    only where AVG does not exist (gloo) or where fewer replicas than ranks ran (the last partial batch of an epoch).
    BatchNorm statistics stay per rank, as they are per replica in the reference.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Variable

from .data import Batch, partition_by_nodes


class DataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, shard_input=True, process_group=None, front_end=None):
        super().__init__()
        self.module = module
        # optional keyword arguments of Batch.from_data_list for the device front-end: knn=(radius, k), mean=, std=
        self.front_end = dict(front_end or {})
        self.shard_input = shard_input
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._flat = None
        # floats of the step sequencer's per-pass gradient buffer when it covers this model (else None): the all-reduce then has
        # that size and layout on every rank, in place where the gradients are already there
        self._seq_total = None
        try:
            from . import native
            total = native.static_flat(module) if next(module.parameters()).is_cuda else None
            if total is not None and sum(len(v) for v in module._flat_index.values()) == len(self._params):
                self._seq_total = total
        except (RuntimeError, OSError, AttributeError, StopIteration):
            self._seq_total = None
        self._pending = False
        self._active = self.world        # ranks that received graphs in the current step (see local_chunk)
        # ReduceOp.AVG exists on the nccl (= RCCL) backend only
        self._has_avg = self.world > 1 and dist.get_backend(process_group) == 'nccl'
        # time_allreduce(True): a HIP event pair on the current stream around every gradient exchange (bench.py's allreduce_ms)
        self._events = None
        if self.world > 1:
            with torch.no_grad():                      # replicas start from rank 0's weights and buffers
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=self.group)
            for p in self._params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- device of the replica
    @property
    def device(self):
        return next(self.module.parameters()).device

    # ---- gradient exchange: queued once per backward pass, runs when the autograd engine has finished
    def _on_grad(self, _param):
        if not self._pending:
            self._pending = True
            Variable._execution_engine.queue_callback(self._allreduce_grads)

    def _step_buffer(self):
        """The ONE buffer the step sequencer left every gradient of this backward pass in (native._grad_buffer), or None when the
        gradients are not (all) there: a pass on the per-operator path, gradients accumulated over several passes, an idle rank."""
        m = self.module
        flat, index, taken = m.__dict__.get('_step_flat'), m._flat_index, getattr(m, '_step_taken', ())
        if flat is None or len(taken) != 4 or flat.numel() != self._seq_total:
            return None
        for slot, items in index.items():
            p, off = items[0]
            g, f = p.grad, m._flat_grads.get(slot)
            if g is None or f is None or g.data_ptr() != f.data_ptr() + 4 * off:
                return None
            lo = f.data_ptr() - flat.data_ptr()
            if lo < 0 or lo + 4 * f.numel() > 4 * flat.numel():
                return None
        return flat

    def time_allreduce(self, enabled=True):
        """Record a HIP event pair around every gradient exchange from now on (on torch's current stream: the collective's own
        stream is joined to it before the call returns).  allreduce_ms() reads them back."""
        self._events = [] if enabled else None
        return self

    def allreduce_ms(self):
        """Durations (ms) of the exchanges recorded since time_allreduce(); synchronises."""
        if not self._events:
            return []
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return out

    def _mean_over_replicas(self, buf):
        """buf <- sum over ranks / active replicas, in place."""
        ev = None
        if self._events is not None and buf.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self._has_avg and self._active == self.world:
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.div_(self._active)      # mean over the replicas that ran (train.py:179 torch.mean(cls_loss))
        if ev is not None:
            ev[1].record()
            self._events.append(ev)

    def _allreduce_grads(self):
        self._pending = False
        if self._seq_total is not None:
            step = self._step_buffer()
            if step is not None:           # in place: no gather into a bucket, no scatter back (padding between the slices rides along)
                self._mean_over_replicas(step)
                return
            # same collective, same layout, from wherever the gradients are (zeros where a parameter has none)
            from . import native
            m = self.module
            dev = self.device
            buf = torch.zeros(self._seq_total, dtype=torch.float32, device=dev)
            views, grads = [], []
            for slot, items in m._flat_index.items():
                base = native.flat_slot_offset(m, slot)
                for p, off in items:
                    if p.grad is not None:
                        views.append(buf[base + off:base + off + p.numel()].view_as(p))
                        grads.append(p.grad)
            if grads:
                torch._foreach_copy_(views, grads)
            self._mean_over_replicas(buf)
            if grads:
                torch._foreach_copy_(grads, views)
            return
        grads = [p.grad for p in self._params if p.grad is not None]
        if not grads:
            return
        total = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != total or self._flat.device != grads[0].device:
            self._flat = torch.empty(total, dtype=torch.float32, device=grads[0].device)
        views, off = [], 0
        for g in grads:
            views.append(self._flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        self._mean_over_replicas(self._flat)
        torch._foreach_copy_(grads, views)

    # ---- forward
    def local_chunk(self, data_list):
        if self.world == 1 or not self.shard_input:
            return data_list
        # torch_geometric's scatter simply uses FEWER devices when the split yields fewer non-empty chunks (the last partial
        # batch of an epoch, one huge graph dominating the node-count split): the ranks beyond them idle through the step
        # -- they still join the gradient all-reduce, contributing zeros -- and the mean is taken over the active replicas
        chunks = partition_by_nodes(data_list, self.world)
        self._active = len(chunks)
        return chunks[self.rank] if self.rank < len(chunks) else []

    def _idle_step(self):
        """This rank got no graphs: empty logits; in training mode a zero loss that still reaches every parameter, so that
        backward fires the hooks and the rank takes part in the all-reduce."""
        out_dim = [m for m in self.module.modules() if isinstance(m, nn.Linear)][-1].out_features
        logits = torch.zeros(0, out_dim, device=self.device)
        if not self.module.training:
            return logits
        return logits, sum(p.sum() for p in self._params) * 0.0

    def forward(self, data_list):
        """data_list: python list of Data (the DataListLoader protocol).  Returns what the module returns
        for this rank's chunk: ``(logits, loss)`` in training mode, ``logits`` in eval mode."""
        if hasattr(data_list, 'edge_index'):          # an already collated (device-resident) Batch of THIS rank
            self._active = self.world
            return self.module(data_list)
        if len(data_list) == 0:
            raise ValueError('empty batch')
        chunk = self.local_chunk(data_list)
        if len(chunk) == 0:
            return self._idle_step()
        # loader front-end on the device: one packed copy + one kernel (data.py / csrc/collate.hip), optionally with the
        # k-NN graph construction and the feature z-scoring the reference does per item on the host
        batch = Batch.from_data_list(chunk, device=self.device, **self.front_end)
        return self.module(batch)
