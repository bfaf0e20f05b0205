"""The training step AROUND the backbone (reference `train.py:174-206,289,360-362`).

The reference's step is: forward, seven loss functions built from ~20 small torch kernels, eight `.item()` host
synchronisations for logging, `loss.backward()`, a 260-tensor AdamW loop, and an exponential LR decay per epoch.  Once the
backbone itself runs in tens of milliseconds those pieces show, so they are part of the hot path here:

  pose_loss(pred, gt, ...)   loss_mpjpe + lambda_scale * n_mpjpe + lambda_velocity * loss_velocity and its gradient in ONE
                             kernel pass (`mbx_pose_loss`); the four loss values stay on the device.
  FlatAdamW                  the model's parameters re-laid into ONE flat fp32 buffer in backward-completion order -- the
                             same order in which the backbone's backward writes its single flat gradient buffer -- so that
                             `step()` is one launch of `mbx_adamw_step` over 42.5 M elements (and, under data parallelism,
                             the all-reduced buckets ARE the optimizer's input, no gather).  Step count and learning rate
                             live on the device.
  GraphedTrainStep           forward + loss + backward + update captured once into a hipGraph and replayed per batch: no
                             Python / ctypes launch cost (~850 launches per step), which is what bounds small batches.
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import grad_bucket
from .model import named_parameter_tensors


class _PoseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, pred, gt, lambda_scale, lambda_velocity):
        pred_c, gt_c = pred.contiguous().float(), gt.contiguous().float()
        losses = torch.empty(4, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred_c) if ctx.needs_input_grad[1] else None
        ops.pose_loss(pred_c, gt_c, lambda_scale, lambda_velocity, losses, dpred)
        ctx.dpred = dpred
        ctx.mark_non_differentiable(losses)
        return losses[3].clone(), losses

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dtotal, _dlosses):
        d = ctx.dpred
        ctx.dpred = None
        return None, (d * dtotal if d is not None else None), None, None, None


def pose_loss(pred: torch.Tensor, gt: torch.Tensor, lambda_scale: float = 0.5, lambda_velocity: float = 20.0, ops=None):
    """`(total, losses)` with `losses = [mpjpe, n_mpjpe, velocity, total]` (device tensor, no host sync) for
    pred, gt [B,T,J,3]: `loss_mpjpe + lambda_scale * n_mpjpe + lambda_velocity * loss_velocity`
    (lib/model/loss.py:56-62,81-91,133-142 combined as train.py:176-189; defaults = configs/pose3d/MB_train_h36m.yaml:36-43).
    `total` is differentiable with respect to `pred`; its gradient was computed in the same kernel pass."""
    if ops is None:
        from . import hip_ops
        ops = hip_ops.get()
    return _PoseLossFn.apply(ops, pred, gt, float(lambda_scale), float(lambda_velocity))


class _Loss2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, pred, target, conf):
        pred_c = pred.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred_c) if ctx.needs_input_grad[1] else None
        ops.loss_2d_weighted(pred_c, target, conf, loss, dpred)
        ctx.dpred = dpred
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss):
        d = ctx.dpred
        ctx.dpred = None
        return None, (d * dloss if d is not None else None), None, None


def loss_2d_weighted(pred: torch.Tensor, target: torch.Tensor, conf: torch.Tensor, ops=None) -> torch.Tensor:
    """`mean(|| (pred[..., :2] - target[..., :2]) * conf ||)` (lib/model/loss.py:72-77; the 2D branch of pre-training,
    train.py:200-203) with its gradient in the same kernel pass.  `target` [B,T,J,>=2] and `conf` [B,T,J,1] may be views of
    the 2D batch itself (`batch_gt`, `batch_input[..., 2:]`): no deep copy of the confidence (train.py:164)."""
    if ops is None:
        from . import hip_ops
        ops = hip_ops.get()
    return _Loss2DFn.apply(ops, pred, target.detach(), conf.detach())


class PretrainStep:
    """One optimizer step of the reference's `train_epoch` (train.py:155-206) with every piece on the device:

        step = PretrainStep(net, optimizer, aug=Augmenter2D(args), rootrel=True, mask=True, noise=True,
                            lambda_scale=0.5, lambda_velocity=20.0)               # configs/pretrain/MB_pretrain.yaml
        losses = step(batch_input, batch_gt, has_3d=True, has_gt=True)          # 3D batches  [B,243,17,3]
        losses = step(batch_2d, batch_2d, has_3d=False, has_gt=...)            # 2D batches  [B,81|30,17,3], target = input

    has_3d: pose losses `loss_mpjpe + lambda_scale * n_mpjpe + lambda_3d_velocity * loss_velocity` (the other lambdas are 0
    in every shipped config and are refused otherwise) -> `losses` = [mpjpe, n_mpjpe, velocity, total];
    not has_3d: `loss_2d_weighted(pred, batch_gt, conf)` with conf = the input's third channel BEFORE augmentation
    (train.py:163-164) -> `losses` = [0, 0, 0, 2d_proj].  `rootrel`: batch_gt -= batch_gt[:, :, 0:1] (train.py:165-166), else the
    depth of the first frame's root is moved to 0 (:168).  `aug.augment2D(batch_input, noise=noise and has_gt, mask=mask)`
    (train.py:169-170) runs as one kernel (motionbert_amd.augment).  No `.item()`: the loss values stay on the device.
    `net` is the backbone or its `DistributedDSTformer` wrapper; under data parallelism every rank must call the step the
    same number of times per loader (`pretrain_epoch_plan`), the gradient all-reduce is the only exchange."""

    def __init__(self, net, optimizer, aug=None, rootrel: bool = True, mask: bool = True, noise: bool = True, no_conf: bool = False,
                 lambda_scale: float = 0.5, lambda_velocity: float = 20.0, lambda_lv=0.0, lambda_lg=0.0, lambda_a=0.0, lambda_av=0.0):
        if any(float(v) != 0.0 for v in (lambda_lv, lambda_lg, lambda_a, lambda_av)):
            raise NotImplementedError('limb / angle losses (lambda_lv, lambda_lg, lambda_a, lambda_av) are 0 in every shipped config; '
                                      'only loss_mpjpe + n_mpjpe + loss_velocity are fused')
        if (mask or noise) and aug is None:
            raise ValueError('mask / noise need an Augmenter2D (motionbert_amd.augment.Augmenter2D(args))')
        self.net, self.opt, self.aug = net, optimizer, aug
        self.rootrel, self.mask, self.noise, self.no_conf = rootrel, mask, noise, no_conf
        self.ls, self.lv = float(lambda_scale), float(lambda_velocity)

    def __call__(self, batch_input: torch.Tensor, batch_gt: torch.Tensor, has_3d: bool, has_gt: bool = True, seed=None) -> torch.Tensor:
        with torch.no_grad():
            conf = None
            if self.no_conf:
                batch_input = batch_input[..., :2]
            if not has_3d:
                conf = batch_input[..., 2:]                      # a view: the augmentation below returns a NEW tensor
            if self.rootrel:
                batch_gt = batch_gt - batch_gt[:, :, 0:1, :]
            else:
                batch_gt = batch_gt.clone()
                batch_gt[..., 2] = batch_gt[..., 2] - batch_gt[:, 0:1, 0:1, 2]
            if self.mask or self.noise:
                batch_input = self.aug.augment2D(batch_input, noise=(self.noise and has_gt), mask=self.mask, seed=seed)
        pred = self.net(batch_input)
        self.opt.zero_grad(set_to_none=True)
        if has_3d:
            total, losses = pose_loss(pred, batch_gt, self.ls, self.lv)
        else:
            total = loss_2d_weighted(pred, batch_gt, conf)
            z = total.detach() * 0
            losses = torch.stack([z, z, z, total.detach()])
        total.backward()
        self.opt.step()
        return losses


def pretrain_epoch_plan(n_posetrack: int, n_instav: int, n_3d: int, epoch: int, train_2d: bool = True, curriculum: int = 30):
    """The loader sequence of one pre-training epoch (train.py:325-330): from epoch `pretrain_3d_curriculum` on, every
    PoseTrack batch, then every InstaVariety batch, then the 3D batches -- [(loader, has_3d, has_gt, n_batches)].  The batch
    counts must be the per-rank counts of equal shards (PackedMotion3D.epoch_indices wraps the tail like DistributedSampler):
    all ranks then issue the same steps in the same order and the gradient all-reduces pair up."""
    plan = []
    if train_2d and epoch >= curriculum:
        plan += [('posetrack', False, True, int(n_posetrack)), ('instav', False, False, int(n_instav))]
    plan.append(('3d', True, True, int(n_3d)))
    return plan


class ActionStep:
    """One optimizer step of train_action.py:172-188: scores = ActionNet(batch [N,M,T,17,3]), cross-entropy, backward, and the
    two AdamW groups of train_action.py:143-149 -- backbone at `lr_backbone`, head at `lr_head`
    (MB_ft_NTU60_xsub.yaml:7-9) -- as two flat one-launch optimizers; `decay()` is the per-epoch StepLR(gamma=lr_decay).
    `distributed=True` (after init_process_group): the backbone is wrapped as `DistributedDSTformer(backbone, extra=model.head)`
    and attached, so that `model(batch)` itself all-reduces the backbone's gradient buckets while backward runs and the head's
    gradients by post-accumulate hooks (they come first in backward); BatchNorm statistics stay per rank as under the
    reference's nn.DataParallel."""

    def __init__(self, model, lr_backbone: float = 1e-4, lr_head: float = 1e-3, weight_decay: float = 0.01, distributed: bool = False,
                 process_group=None, ops=None):
        self.model, self.ddp = model, None
        if distributed:
            from .ddp import DistributedDSTformer
            self.ddp = DistributedDSTformer(model.backbone, process_group=process_group, extra=model.head, ops=ops).attach()
        self.opt_backbone = FlatAdamW(model.backbone, lr=lr_backbone, weight_decay=weight_decay)
        self.opt_head = FlatAdamW([('head.' + n, p) for n, p in model.head.named_parameters() if p.requires_grad], lr=lr_head,
                                  weight_decay=weight_decay)

    def __call__(self, batch_input: torch.Tensor, labels: torch.Tensor):
        out = self.model(batch_input)
        self.opt_backbone.zero_grad(set_to_none=True)
        self.opt_head.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(out, labels)
        loss.backward()
        if self.ddp is not None:
            self.ddp.wait()                      # (the backbone's backward already waited; covers a frozen backbone)
        self.opt_backbone.step()
        self.opt_head.step()
        return loss.detach(), out.detach()

    def decay(self, gamma: float):
        self.opt_backbone.lr = self.opt_backbone.lr * gamma
        self.opt_head.lr = self.opt_head.lr * gamma


def flat_layout(names, shapes, depth):
    """Offsets of every parameter in the flat buffer, in backward-completion order (tail | levels last..first | embedding):
    the layout `_DSTformerFn.backward` uses for the gradients."""
    order = sorted(range(len(names)), key=lambda i: (grad_bucket(names[i], depth), i))
    offs, off = {}, 0
    for i in order:
        n = int(torch.Size(shapes[i]).numel())
        offs[names[i]] = (off, n)
        off += n
    return offs, off


class FlatAdamW(torch.optim.Optimizer):
    """AdamW (torch.optim.AdamW semantics; reference train.py:289 `optim.AdamW(..., lr, weight_decay)`) over ONE flat fp32
    parameter buffer: one kernel launch per step for everything that has a gradient.

        opt = FlatAdamW(model, lr=2e-4, weight_decay=0.01)       # re-lays model parameters into one buffer (values kept)
        loss.backward(); opt.step(); opt.zero_grad()
        opt.lr = opt.lr * 0.99                                     # train.py:360-362: lr *= lr_decay each epoch

    `model` is a `motionbert_amd.DSTformer` (flat layout = backward-completion order, the order in which the backbone's
    backward writes its single flat gradient buffer: `step()` then uses that buffer as it is) or any other `nn.Module` /
    list of `(name, parameter)` pairs (plain layout, gradients packed first -- e.g. the ActionNet head with its own
    learning rate, MB_ft_NTU60_xsub.yaml:7-9).

    Like torch.optim.AdamW, a parameter WITHOUT a gradient is skipped entirely -- no weight decay, no moment update: frozen
    layers (`partial_train`, learning.py:69-77; the reference builds its optimizer over `requires_grad` parameters only,
    train.py:284-289) and the backbone's unused `head.*` under `get_representation` keep their values bit for bit.  The
    flat buffer is then updated range by range (one launch per contiguous run of parameters that have gradients).

    Deviation from torch.optim.AdamW, by design of the one-launch form: ONE step counter for the whole buffer (kept on the device).
    A parameter that receives its first gradient later than the others (a layer unfrozen mid-training, `head.*` after
    `get_representation`-only steps) gets the bias correction of the GLOBAL step, not of its own first step; `state_dict()`
    therefore writes the global step (and zero moments) for parameters that were never updated, and `load_state_dict()` sets the
    counter to the largest per-parameter step it finds.  The reference never unfreezes mid-run (learning.py:69-77 decides once,
    before the optimizer is built), where the two coincide."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ops=None):
        if isinstance(model, torch.nn.Module):
            names, params = named_parameter_tensors(model) if hasattr(model, '_param_names') else zip(*model.named_parameters())
        else:
            names, params = zip(*list(model))
        if not params or any(not p.is_cuda for p in params):
            raise RuntimeError('FlatAdamW needs the parameters on a ROCm device (move the model first: model.cuda())')
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._names, self._params = list(names), list(params)
        self._backbone = isinstance(model, torch.nn.Module) and hasattr(model, '_param_names') and hasattr(model, 'depth')
        if self._backbone:
            self._offs, total = flat_layout(names, [p.shape for p in params], model.depth)
        else:
            self._offs, total = {}, 0
            for n, p in zip(names, params):
                self._offs[n] = (total, p.numel())
                total += p.numel()
        self._total, self._n = total, (total + 3) // 4 * 4
        self._order = sorted(range(len(self._names)), key=lambda i: self._offs[self._names[i]][0])      # parameters in flat order
        dev = params[0].device
        self.flat = torch.zeros(self._n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p in zip(names, params):
                o, k = self._offs[n]
                self.flat[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + k].view(p.shape)          # the module's parameters now ARE the flat buffer
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.state_t = torch.tensor([0.0, float(lr)], dtype=torch.float32, device=dev)     # {step, lr} on the device
        self._lr = float(lr)
        self._gpack = None
        self._ops = ops

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, v: float):
        self._lr = float(v)
        self.state_t[1] = float(v)          # device scalar: picked up by a captured graph without re-capture
        for g in self.param_groups:
            g['lr'] = float(v)

    def _active_ranges(self):
        """[(lo, hi)] runs of the flat buffer covered by parameters that HAVE a gradient (merged when adjacent)."""
        runs = []
        for i in self._order:
            if self._params[i].grad is None:
                continue
            o, k = self._offs[self._names[i]]
            if runs and runs[-1][1] == o:
                runs[-1][1] = o + k
            else:
                runs.append([o, o + k])
        return [(a, b) for a, b in runs if b > a]

    def _flat_grad(self):
        """The gradients as one flat tensor in the optimizer's order.  One backward pass of the backbone hands every parameter
        a VIEW of a single flat buffer laid out exactly like `self.flat` (model._DSTformerFn.backward): then that buffer is
        used as it is (parameters it did not differentiate -- `head.*` on the representation path -- have no gradient and are
        not part of any range).  Anything else (accumulated or foreign gradients, another layout) is packed into a scratch buffer."""
        first = next((self._params[i].grad for i in self._order if self._params[i].grad is not None), None)
        if first is None:
            return None
        if self._backbone and first.dtype == torch.float32:
            st = first.untyped_storage()
            ok = st.nbytes() >= 4 * self._n
            if ok:
                for n, p in zip(self._names, self._params):
                    g = p.grad                       # (AccumulateGrad keeps the handed-over view, detached: same storage)
                    if g is None:
                        continue
                    if (g.dtype != torch.float32 or not g.is_contiguous() or g.storage_offset() != self._offs[n][0]
                            or g.untyped_storage().data_ptr() != st.data_ptr()):
                        ok = False
                        break
            if ok:
                return torch.empty(0, dtype=torch.float32, device=first.device).set_(st, 0, (self._n,), (1,))
        if self._gpack is None:
            self._gpack = torch.zeros_like(self.flat)
        for n, p in zip(self._names, self._params):
            if p.grad is not None:
                o, k = self._offs[n]
                self._gpack[o:o + k].copy_(p.grad.reshape(-1))
        return self._gpack

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ops = self._ops
        if ops is None:
            from . import hip_ops
            ops = hip_ops.get()
        grp = self.param_groups[0]
        if grp['lr'] != self._lr:              # someone (an LR scheduler) wrote param_groups directly
            self.lr = grp['lr']
        g = self._flat_grad()
        if g is None:                          # nothing has a gradient: torch.optim.AdamW does nothing either (no step count)
            return loss
        runs = self._active_ranges()
        if runs == [(0, self._total)]:
            runs = [(0, self._n)]              # everything active: the padded buffer in one aligned launch
        with torch.cuda.device(self.flat.device):
            for k, (lo, hi) in enumerate(runs):
                ops.adamw_step(self.flat[lo:hi], g[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.state_t, grp['betas'][0],
                               grp['betas'][1], grp['eps'], grp['weight_decay'], tick=(k == 0))
        for p in self._params:                 # the kernel wrote through raw pointers: tell autograd the parameters changed
            if p.grad is not None:
                torch.autograd.graph.increment_version(p)
        return loss

    # ---- checkpointing: torch.optim.AdamW's own format ({'state': {index: {step, exp_avg, exp_avg_sq}}, 'param_groups': [...]}),
    # parameter index = position in model.parameters() -- what the reference saves with `optimizer.state_dict()` (train.py:
    # save_checkpoint) and reloads on resume, so its checkpoints load here and vice versa.  Tensors are copies.
    def state_dict(self):
        step = self.state_t[0].detach().clone()
        state = {}
        if float(step) > 0:
            for i, (n, p) in enumerate(zip(self._names, self._params)):
                o, k = self._offs[n]
                state[i] = dict(step=step.clone().cpu(), exp_avg=self.exp_avg[o:o + k].view(p.shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[o:o + k].view(p.shape).clone())
        # (the keys torch.optim.AdamW itself keeps in a group, so that the dict also loads into a torch optimizer)
        grp = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        grp.update({k: v for k, v in self.param_groups[0].items() if k != 'params'})
        grp['lr'] = self._lr
        grp['params'] = list(range(len(self._params)))
        return dict(state=state, param_groups=[grp], names=list(self._names))

    def load_state_dict(self, sd):
        if 'state' not in sd or 'param_groups' not in sd:
            raise KeyError("FlatAdamW.load_state_dict expects torch.optim.AdamW's format: {'state': {...}, 'param_groups': [...]}")
        if 'names' in sd and list(sd['names']) != self._names:
            raise ValueError('optimizer state was saved for a different parameter list (names differ)')
        ids = [i for g in sd['param_groups'] for i in g['params']]
        if len(ids) != len(self._params):
            raise ValueError(f'optimizer state has {len(ids)} parameters, this model has {len(self._params)}')
        pos = {pid: k for k, pid in enumerate(ids)}          # saved parameter id -> position in model.parameters()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        step = 0.0
        for pid, st in sd['state'].items():
            k = pos[int(pid)]
            n, p = self._names[k], self._params[k]
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state of {n}: shape {tuple(st["exp_avg"].shape)} != parameter {tuple(p.shape)}')
            o, cnt = self._offs[n]
            self.exp_avg[o:o + cnt].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + cnt].copy_(st['exp_avg_sq'].reshape(-1))
            step = max(step, float(st['step']))
        g0 = sd['param_groups'][0]
        for k in ('betas', 'eps', 'weight_decay'):
            if k in g0:
                self.param_groups[0][k] = tuple(g0[k]) if k == 'betas' else g0[k]
        self.state_t[0] = step
        self.lr = float(g0.get('lr', self._lr))


class GraphedTrainStep:
    """forward + fused pose loss + backward + FlatAdamW update as ONE hipGraph, replayed per batch (single GPU).

        step = GraphedTrainStep(model, opt, x_example, gt_example)        # captures; the example batch is NOT trained on
        losses = step(x, gt)            # device tensor [mpjpe, n_mpjpe, velocity, total] of THIS batch; no host sync

    Everything the step reads per batch (`x`, `gt`, the learning rate, the AdamW step count) lives in device memory that the
    graph re-reads at replay, so `opt.lr = ...` needs no re-capture; a new batch SHAPE does."""

    def __init__(self, model, optimizer: FlatAdamW, x: torch.Tensor, gt: torch.Tensor, lambda_scale: float = 0.5,
                 lambda_velocity: float = 20.0, warmup: int = 2):
        if not x.is_cuda:
            raise RuntimeError('GraphedTrainStep needs ROCm device tensors')
        if any(r > 0 for r in getattr(model, 'drop_rates', ())):
            # the dropout / DropPath base seed is a host integer drawn per forward (model.run) and passed to the kernels by value:
            # a captured graph would replay ONE mask forever
            raise NotImplementedError('GraphedTrainStep: dropout / DropPath rates > 0 are not supported under graph capture '
                                      '(the mask seed would be baked into the graph); train eagerly or set the rates to 0')
        self.model, self.opt = model, optimizer
        self.ls, self.lv = float(lambda_scale), float(lambda_velocity)
        self.x, self.gt = x.detach().clone().contiguous().float(), gt.detach().clone().contiguous().float()
        model.train()
        # warm-up on a side stream (allocator pools, descriptor caches, lazy kernel loading) WITHOUT changing the training
        # state: parameters, moments and the step count are restored afterwards
        keep = (optimizer.flat.clone(), optimizer.exp_avg.clone(), optimizer.exp_avg_sq.clone(), optimizer.state_t.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._one()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.losses = self._one()
        torch.cuda.synchronize()
        with torch.no_grad():
            optimizer.flat.copy_(keep[0]); optimizer.exp_avg.copy_(keep[1]); optimizer.exp_avg_sq.copy_(keep[2]); optimizer.state_t.copy_(keep[3])

    def _one(self):
        self.opt.zero_grad(set_to_none=True)
        total, losses = pose_loss(self.model(self.x), self.gt, self.ls, self.lv)
        total.backward()
        self.opt.step()
        return losses

    def __call__(self, x: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        if tuple(x.shape) != tuple(self.x.shape) or tuple(gt.shape) != tuple(self.gt.shape):
            raise ValueError(f'graph was captured for {tuple(self.x.shape)} / {tuple(self.gt.shape)}, got {tuple(x.shape)} / {tuple(gt.shape)}')
        self.x.copy_(x)
        self.gt.copy_(gt)
        self.graph.replay()
        return self.losses.clone()
