"""Data-parallel training of the DSTformer hot path: one process per GPU, gradients only over RCCL.

The reference's only multi-GPU mechanism is single-process `nn.DataParallel` (train.py:256-258:
per-step parameter broadcast + output gather through GPU 0).  Here every rank owns a full replica
and its shard of the minibatch (clips are independent: no cross-sample op in the backbone), and the
only exchange is the gradient all-reduce.  Because the whole backbone is ONE autograd node whose
backward is a fixed kernel sequence, the overlap is organised by the engine itself rather than by
per-parameter hooks: the flat fp32 gradient buffer is laid out in backward completion order and cut
into depth + 2 buckets (tail | level depth-1 | ... | level 0 | embedding; ~33.6 MB per level for
the full model).  As soon as the last kernel writing a bucket has been enqueued, that bucket is
all-reduced as one contiguous RCCL call on the communication stream while the next level's backward
keeps the compute stream busy; `finish()` makes the compute stream wait for the outstanding
collectives before autograd hands the gradients out.  xGMI is point-to-point (7 links per GPU), so
few large ring all-reduces beat many small ones: 7 calls per step, none below 2 MB.

    model = DSTformer(...).cuda()
    ddp = DistributedDSTformer(model)          # after torch.distributed.init_process_group('nccl')
    loss = criterion(ddp(x_local), y_local); loss.backward(); optimizer.step()

Works with any `torch.distributed` backend (the CPU tests run it over gloo with world_size 2).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


def _mean_all_reduce(t: torch.Tensor, group, world: int):
    """Asynchronous mean over the ranks.  RCCL ('nccl') averages inside the collective (ReduceOp.AVG: no extra pass over the
    bucket); gloo has no AVG, there the tensor is pre-divided so that SUM yields the mean."""
    if dist.get_backend(group) == 'nccl':
        return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=True)
    t.div_(world)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


class _BucketSync:
    """Receives finished gradient buckets from the backward pass and all-reduces them asynchronously."""

    def __init__(self, group=None, pending=None, diag=None):
        self.group, self.world = group, dist.get_world_size(group)
        self.pending = pending if pending is not None else []     # shared with the wrapper's extra-parameter hooks
        self.diag = diag                                           # optional dict (DistributedDSTformer.diagnostics)

    def bucket_ready(self, flat_view: torch.Tensor):
        if flat_view.numel() == 0:
            return
        if self.diag is not None:
            self.diag.setdefault('bucket_bytes', []).append(flat_view.numel() * flat_view.element_size())
        self.pending.append(_mean_all_reduce(flat_view, self.group, self.world))

    def finish(self):
        """The compute stream waits here for the outstanding collectives.  With diagnostics on, the wait is bracketed by two events
        on that stream: their distance is the part of the all-reduce time NOT hidden under backward ('exposed')."""
        ev = None
        if self.diag is not None and torch.cuda.is_available() and self.pending:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        while self.pending:
            self.pending.pop(0).wait()          # stream-level wait on GPU backends; blocks on gloo
        if ev is not None:
            ev[1].record()
            self.diag.setdefault('wait_events', []).append(ev)


class DistributedDSTformer(nn.Module):
    """Gradient-averaging wrapper around a `motionbert_amd.DSTformer` replica (one per process/GPU)."""

    def __init__(self, module: nn.Module, process_group=None, broadcast_parameters: bool = True, ops=None, extra=None):
        """`extra`: a module (or an iterable of parameters) that lives OUTSIDE the backbone but trains with it -- the
        ActionNet head `fc1 / bn / fc2` of `lib/model/model_action.py:15-29`, a mesh regressor ...  Their parameters (and
        buffers, e.g. BatchNorm running statistics) are broadcast from rank 0 like the backbone's, and each of their
        gradients is mean-all-reduced by a post-accumulate hook as soon as autograd has produced it.  A head sits behind the
        backbone in forward, so its gradients come FIRST in backward and their collectives overlap the whole backbone
        backward; `finish()` of the backbone's bucket sync (or `wait()`) waits for them too.  BatchNorm statistics stay
        per rank, as under the reference's nn.DataParallel (SURVEY.md 8e gotcha 2)."""
        super().__init__()
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('DistributedDSTformer needs torch.distributed.init_process_group() first')
        self.module, self.group, self._ops = module, process_group, ops
        self._pending = []
        self.diagnostics = None      # set to a dict to collect per-bucket bytes and the exposed wait of every backward (bench.py --gpus N)
        extra_params, extra_bufs = [], []
        if extra is not None:
            if isinstance(extra, nn.Module):
                inside = {id(p) for p in module.parameters()}
                extra_params = [p for p in extra.parameters() if id(p) not in inside]
                inside_b = {id(b) for b in module.buffers()}
                extra_bufs = [b for b in extra.buffers() if id(b) not in inside_b]
            else:
                extra_params = list(extra)
        self.extra_parameters = extra_params
        world = dist.get_world_size(process_group)
        for p in extra_params:
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(
                    lambda q, g=process_group, w=world: self._pending.append(_mean_all_reduce(q.grad, g, w)))
        if broadcast_parameters:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()) + extra_params + extra_bufs:
                    dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                   group=process_group)

    def attach(self):
        """Make every gradient-enabled forward of the WRAPPED module itself data-parallel (`module(x)`,
        `module.get_representation(x)`, `module.get_pooled_representation(...)`): models that own the backbone as a sub-module
        and call it directly -- `ActionNet.backbone` (model_action.py:62-70), a mesh regressor -- then need no change.  Returns
        self.  `detach()` undoes it."""
        self.module._grad_sync = _BucketSync(self.group, self._pending)
        return self

    def detach(self):
        if hasattr(self.module, '_grad_sync'):
            del self.module._grad_sync

    def wait(self):
        """Block (stream-level on GPU backends) until every outstanding gradient collective has finished.  Called by the
        backbone's backward; call it yourself before optimizer.step() only if the backbone took no part in backward."""
        while self._pending:
            self._pending.pop(0).wait()

    def forward(self, x, return_rep: bool = False):
        from . import model as M
        sync = _BucketSync(self.group, self._pending, self.diagnostics) if torch.is_grad_enabled() else None
        if self._ops is None:
            self.module._check(x)
            from . import hip_ops
            ops = hip_ops.get()
        else:
            ops = self._ops   # explicit kernel provider (tests)
        return M.run(ops, self.module, x.contiguous().float(), return_rep, sync)

    def get_representation(self, x):
        return self.forward(x, return_rep=True)
