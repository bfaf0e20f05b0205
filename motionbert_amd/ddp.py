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


class _BucketSync:
    """Receives finished gradient buckets from the backward pass and all-reduces them asynchronously."""

    def __init__(self, group=None):
        self.group, self.world, self.pending = group, dist.get_world_size(group), []

    def bucket_ready(self, flat_view: torch.Tensor):
        if flat_view.numel() == 0:
            return
        # pre-divide so that SUM yields the mean (ReduceOp.AVG is not available on every backend)
        flat_view.div_(self.world)
        self.pending.append(dist.all_reduce(flat_view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()          # stream-level wait on GPU backends; blocks on gloo
        self.pending = []


class DistributedDSTformer(nn.Module):
    """Gradient-averaging wrapper around a `motionbert_amd.DSTformer` replica (one per process/GPU)."""

    def __init__(self, module: nn.Module, process_group=None, broadcast_parameters: bool = True, ops=None):
        super().__init__()
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('DistributedDSTformer needs torch.distributed.init_process_group() first')
        self.module, self.group, self._ops = module, process_group, ops
        if broadcast_parameters:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                   group=process_group)

    def forward(self, x, return_rep: bool = False):
        from . import model as M
        sync = _BucketSync(self.group) if torch.is_grad_enabled() else None
        if self._ops is None:
            self.module._check(x)
            from . import hip_ops
            ops = hip_ops.get()
        else:
            ops = self._ops   # explicit kernel provider (tests)
        return M.run(ops, self.module, x.contiguous().float(), return_rep, sync)

    def get_representation(self, x):
        return self.forward(x, return_rep=True)
