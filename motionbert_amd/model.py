"""Drop-in replacement for `lib.model.DSTformer.DSTformer` backed by HIP kernels.

Boundary: the reference exposes the hot path as a Python class;
its callers (`lib/utils/learning.py:79-85` load_backbone, `train.py:174`,
`infer_wild.py:75-80`, `lib/model/model_action.py:68`) construct it, call
`model(x)` / `model.get_representation(x)`, wrap it in `nn.DataParallel` and
`load_state_dict(strict=True)` into it.  This class keeps

  * the constructor signature and defaults of `DSTformer.py:270-273`,
  * `forward(x, return_rep=False)` (`:329`), `get_representation` (`:360`),
    `get_classifier` / `reset_classifier` (`:322-327`),
  * the exact parameter tree, i.e. the 260 `state_dict` keys, shapes and order,
  * the initialisation (same RNG consumption order, so the same seed yields
    the same weights as the reference),

and replaces the arithmetic: forward and backward run as hand-written gfx950
kernels reached through the C ABI of `libmbx.so` (see `include/mbx.h`).
The sub-modules below (`MLP`, `Attention`, `Block`) are parameter containers
only; they carry no torch arithmetic and cannot be called.  There is no CPU
path: calling the model on a non-ROCm tensor raises.
"""
from __future__ import annotations

import contextlib
import os
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn as nn

from .engine import Engine, ModelCfg, grad_bucket

#: precision -> dtype of the T-typed tensors.  'bf16x3' computes every Linear as three bf16 MFMA passes over hi / lo operand
#: planes (fp32-class accuracy at a third of the bf16 rate); all its tensors are fp32 like 'fp32'.
_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32, 'bf16x3': torch.float32}


def _trunc_normal_(t, std):
    # same sampling recipe as the reference helper (DSTformer.py:12-66): inverse-CDF
    # truncated normal on [-2, 2]; torch ships the identical routine.
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class _Holder(nn.Module):
    """Parameter container; the fused HIP path does the math."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f'{type(self).__name__} is a parameter container of the fused HIP DSTformer; '
                           'call the DSTformer module instead')


class MLP(_Holder):
    """fc1 -> GELU(erf) -> fc2  (DSTformer.py:69-85)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(_Holder):
    """qkv + proj of one attention sub-layer (DSTformer.py:88-107).  Only the 'spatial'
    and 'temporal' modes are ever built by Block (`:223-226`)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., st_mode='vanilla'):
        super().__init__()
        if st_mode not in ('spatial', 'temporal'):
            raise NotImplementedError(f"Attention mode '{st_mode}' is dead code in the reference and not part of the hot path")
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)       # registered before qkv, as in the reference (RNG order)
        self.mode = st_mode
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(_Holder):
    """Four pre-norm residual sub-layers (DSTformer.py:214-249)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., mlp_out_ratio=1., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, st_mode='stage_st', att_fuse=False):
        super().__init__()
        if st_mode not in ('stage_st', 'stage_ts'):
            raise NotImplementedError(f"Block mode '{st_mode}' is never instantiated by DSTformer")
        if att_fuse or mlp_out_ratio != 1.:
            raise NotImplementedError('Block.att_fuse / mlp_out_ratio are unused by DSTformer')
        self.st_mode = st_mode
        self.norm1_s = norm_layer(dim)
        self.norm1_t = norm_layer(dim)
        self.attn_s = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                attn_drop=attn_drop, proj_drop=drop, st_mode='spatial')
        self.attn_t = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                attn_drop=attn_drop, proj_drop=drop, st_mode='temporal')
        self.drop_path = nn.Identity()
        self.drop_path_rate = float(drop_path)
        self.norm2_s = norm_layer(dim)
        self.norm2_t = norm_layer(dim)
        hidden = int(dim * mlp_ratio)
        self.mlp_s = MLP(in_features=dim, hidden_features=hidden, out_features=dim, act_layer=act_layer, drop=drop)
        self.mlp_t = MLP(in_features=dim, hidden_features=hidden, out_features=dim, act_layer=act_layer, drop=drop)
        self.att_fuse = att_fuse


class _DSTformerFn(torch.autograd.Function):
    """One autograd node for the whole backbone: forward and backward are each a
    fixed sequence of HIP kernel launches on the current stream."""

    @staticmethod
    def forward(ctx, ops, cfg, names, precision, return_rep, grad_sync, x, *params):
        precision, drop_seed = precision if isinstance(precision, tuple) else (precision, None)
        # needs_input_grad ignores torch.no_grad(): `grad_sync` is (enabled, sync) decided by the caller, where
        # the grad mode is still visible (autograd switches it off inside Function.forward)
        grad_enabled, grad_sync = grad_sync
        need_grad = grad_enabled and any(ctx.needs_input_grad[6:])
        P = dict(zip(names, params))
        precision, gelu_d = (precision[:-3], False) if precision.endswith('+nd') else (precision, True)
        precision, fold = (precision[:-3], False) if precision.endswith('+nf') else (precision, True)
        precision, recompute = (precision[:-2], True) if precision.endswith('+r') else (precision, False)
        eng = Engine(ops, cfg, P, _DTYPES[precision], x3=precision == 'bf16x3', drop_seed=drop_seed)
        eng.recompute = recompute
        eng.fold = eng.fold and fold
        eng.gelu_d = eng.gelu_d and gelu_d
        with _device_of(x):
            tta = None
            if isinstance(return_rep, tuple) and return_rep[0] == 'tta':
                tta, return_rep = return_rep[1], False
            out, saved = eng.forward(x, return_rep, need_grad, tta_perm=tta)
        if need_grad:
            ctx.eng, ctx.saved_acts, ctx.names, ctx.grad_sync = eng, saved, names, grad_sync
            ctx.pshapes = [p.shape for p in params]
            ctx.return_rep = bool(return_rep)
            # The kernels of backward read the parameters and (on the representation path) the returned tensor
            # through raw pointers: registering them lets autograd's version counters catch an in-place edit
            # (optimizer.step() before backward, rep.mul_()) instead of silently using the modified bytes.
            ctx.save_for_backward(*params, *((out,) if return_rep is True else ()))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        eng, saved, names = ctx.eng, ctx.saved_acts, ctx.names
        if saved is None:
            raise RuntimeError('DSTformer backward called twice (activations were released)')
        ctx.saved_tensors   # version check only (raises if a parameter / the representation was modified in place)
        dout = dout.contiguous().float()
        # one flat fp32 gradient buffer, laid out in backward completion order (bucket by bucket) so that a
        # data-parallel wrapper can all-reduce each bucket as one contiguous RCCL call while backward runs on
        depth = eng.cfg.depth
        order = sorted(range(len(names)), key=lambda i: (grad_bucket(names[i], depth), i))
        sizes = [int(torch.Size(s).numel()) for s in ctx.pshapes]
        # (padded to a multiple of 4 elements: the flat AdamW kernel of motionbert_amd.train reads it with 16-byte loads)
        total = sum(sizes)
        flat = torch.empty((total + 3) // 4 * 4, dtype=torch.float32, device=dout.device)
        if flat.numel() > total:
            flat[total:].zero_()
        grads: Dict[str, torch.Tensor] = {}
        bounds = [0] * (depth + 3)
        off = 0
        for i in order:
            grads[names[i]] = flat[off:off + sizes[i]].view(ctx.pshapes[i])
            off += sizes[i]
            bounds[grad_bucket(names[i], depth) + 1] = off
        for b in range(1, len(bounds)):          # empty buckets (e.g. no ts_attn) inherit the running offset
            bounds[b] = max(bounds[b], bounds[b - 1])
        sync = ctx.grad_sync
        on_ready = (lambda b: sync.bucket_ready(flat[bounds[b]:bounds[b + 1]])) if sync is not None else None
        with _device_of(dout):
            dx = eng.backward(saved, dout, grads, want_dx=ctx.needs_input_grad[6], on_ready=on_ready)
            if sync is not None:
                sync.finish()
        ctx.saved_acts = None
        ctx.eng = None
        eng.grads = None      # the views handed to autograd below must be the ONLY references (AccumulateGrad then keeps
                              # them as .grad instead of cloning, and FlatAdamW can use the flat buffer as it is)
        # get_representation() never touches the head (DSTformer.py:354-356 returns before :357): like the reference's
        # autograd, hand back NO gradient for it (a zero tensor would let AdamW's weight decay shrink the unused head)
        unused = ('head.weight', 'head.bias') if ctx.return_rep else ()
        gp = tuple(grads.pop(n) if ng and n not in unused else None for n, ng in zip(names, ctx.needs_input_grad[7:]))
        grads.clear()
        del flat
        return (None, None, None, None, None, None, dx) + gp


def make_cfg(model) -> ModelCfg:
    return ModelCfg(dim_in=model.dim_in, dim_out=model.dim_out, C=model.dim_feat, R=model.dim_rep, depth=model.depth,
                    H=model.num_heads, hidden=int(model.dim_feat * model.mlp_ratio), J=model.num_joints,
                    maxlen=model.maxlen, eps=model.ln_eps, scale=model.qk_scale or (model.dim_feat // model.num_heads) ** -0.5,
                    att_fuse=model.att_fuse, qkv_bias=model.qkv_bias, drop=model.drop_rates[0], attn_drop=model.drop_rates[1],
                    dpr=tuple(float(v) for v in torch.linspace(0, model.drop_rates[2], model.depth)))


def _device_of(t):
    """Device guard: every libmbx call enqueues on torch's CURRENT stream, which belongs to the current device --
    make that the device the tensors live on (model.to('cuda:1') with cuda:0 current; autograd / DataParallel threads)."""
    return torch.cuda.device(t.device) if t.is_cuda else contextlib.nullcontext()


def named_parameter_tensors(model):
    """(names, tensors) of the 260 parameters in state_dict order.  On a real module this is named_parameters();
    on an `nn.DataParallel` replica `_parameters` is empty (torch.nn.parallel.replicate attaches the broadcast
    copies as plain attributes), so the tensors are resolved by walking the recorded names with getattr."""
    pairs = list(model.named_parameters())
    if pairs:
        names, params = zip(*pairs)
        return names, params
    names = tuple(model._param_names)
    params = []
    for n in names:
        obj = model
        for part in n.split('.'):
            obj = getattr(obj, part)
        params.append(obj)
    return names, tuple(params)


def run(ops, model, x, return_rep=False, grad_sync=None):
    """Run the fused path of `model` on `x` with an explicit kernel provider.  `grad_sync` (optional) is
    told about finished gradient buckets during backward (see motionbert_amd.ddp)."""
    cfg = make_cfg(model)
    names, params = named_parameter_tensors(model)
    for n, p in zip(names, params):
        if p.device != x.device or p.dtype != torch.float32:
            raise RuntimeError(f'parameter {n} is {p.dtype} on {p.device} but the input is on {x.device}: the HIP path needs '
                               'fp32 parameters on the input\'s device (model.to(x.device))')
    drop_seed = None
    if model.training and any(r > 0 for r in model.drop_rates):
        # base seed of this forward's dropout masks: the model's own override (tests) or torch's global CPU generator
        drop_seed = getattr(model, '_drop_seed', None)
        if drop_seed is None:
            drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _DSTformerFn.apply(ops, cfg, names, (model.precision + ('+r' if getattr(model, 'recompute', False) else '') + ('' if getattr(model, 'fold_ln', True) else '+nf') + ('' if getattr(model, 'gelu_d', True) else '+nd'), drop_seed), return_rep, (torch.is_grad_enabled(), grad_sync), x, *params)


class DSTformer(nn.Module):
    def __init__(self, dim_in=3, dim_out=3, dim_feat=256, dim_rep=512,
                 depth=5, num_heads=8, mlp_ratio=4,
                 num_joints=17, maxlen=243,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, att_fuse=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_feat, self.dim_rep = dim_in, dim_out, dim_feat, dim_rep
        self.depth, self.num_heads, self.mlp_ratio = depth, num_heads, mlp_ratio
        self.num_joints, self.maxlen = num_joints, maxlen
        self.qkv_bias, self.qk_scale = qkv_bias, qk_scale
        self.drop_rates = (float(drop_rate), float(attn_drop_rate), float(drop_path_rate))
        #: arithmetic of the GEMM/attention operands: 'bf16' (MFMA bf16, fp32 accumulate, fp32 residual stream and
        #: statistics; the throughput mode), 'bf16x3' (split-operand bf16 MFMA, fp32-class: meets the 1e-3 gate at about a
        #: third of the bf16 GEMM rate) or 'fp32' (exact fp32 MFMA; the reference parity mode)
        self.precision = os.environ.get('MBX_PRECISION', 'bf16')
        #: low-memory training: rebuild the MLP post-activations (and, without LayerNorm folding, the LayerNorm outputs) in backward
        #: instead of saving them (one extra element-wise launch each; 256 clips x 243 frames then train inside one MI355X)
        self.recompute = False
        #: LayerNorm folding (bf16, no dropout): the affine part of every Block LayerNorm lives in the qkv / fc1 weights and its
        #: backward runs as the epilogue of the dX GEMM (engine.py, include/mbx.h).  False selects the plain sequencing (the A/B
        #: switch of the measurements; the fp32-class modes and training with dropout use the plain sequencing anyway).
        self.fold_ln = True
        #: bf16 training: fc1's epilogue saves gelu'(u) (from the fp32 accumulator) instead of the pre-activation wherever the row-owner
        #: LayerNorm-backward GEMM follows in backward; False keeps the pre-activation and the GELU' epilogue (the A/B switch, and what
        #: `recompute` uses anyway).  The two forward epilogues use different erf approximations (both ~1e-7): outputs agree to bf16
        #: rounding, not bit for bit.
        self.gelu_d = True
        self.joints_embed = nn.Linear(dim_in, dim_feat)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        mk = lambda i, mode: Block(dim=dim_feat, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                   qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i],
                                   norm_layer=norm_layer, st_mode=mode)
        self.blocks_st = nn.ModuleList([mk(i, 'stage_st') for i in range(depth)])
        self.blocks_ts = nn.ModuleList([mk(i, 'stage_ts') for i in range(depth)])
        self.norm = norm_layer(dim_feat)
        if not isinstance(self.norm, nn.LayerNorm) or not self.norm.elementwise_affine:
            raise NotImplementedError('norm_layer must build an affine nn.LayerNorm (learning.py:84 passes partial(nn.LayerNorm, eps=1e-6))')
        self.ln_eps = float(self.norm.eps)
        if dim_rep:
            self.pre_logits = nn.Sequential(OrderedDict([('fc', nn.Linear(dim_feat, dim_rep)), ('act', nn.Tanh())]))
        else:
            self.pre_logits = nn.Identity()
        self.head = nn.Linear(dim_rep, dim_out) if dim_out > 0 else nn.Identity()
        self.temp_embed = nn.Parameter(torch.zeros(1, maxlen, 1, dim_feat))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_joints, dim_feat))
        _trunc_normal_(self.temp_embed, std=.02)
        _trunc_normal_(self.pos_embed, std=.02)
        self.apply(self._init_weights)
        self.att_fuse = att_fuse
        if self.att_fuse:
            # built after apply(): starts as an even blend (weight 0, bias 0.5), DSTformer.py:307-311
            self.ts_attn = nn.ModuleList([nn.Linear(dim_feat * 2, 2) for _ in range(depth)])
            for lin in self.ts_attn:
                lin.weight.data.fill_(0)
                lin.bias.data.fill_(0.5)
        self._param_names = [n for n, _ in self.named_parameters()]   # for DataParallel replicas, see named_parameter_tensors

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ prepared-weight cache of the no-grad path (engine.prepare_weights)
    @staticmethod
    def _drop_weight_cache():
        from . import hip_ops
        ops = hip_ops.peek()
        if ops is not None:
            ops.weight_cache.clear()

    def train(self, mode: bool = True):
        # entering (or leaving) training: the folded / packed weights kept for inference calls are about to go stale -- also for
        # updates the version counters do not see (`p.data.add_()`, EMA copies, a replayed hipGraph step): ADVICE r5
        if mode != self.training:
            self._drop_weight_cache()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._drop_weight_cache()
        return out

    def get_classifier(self):
        return self.head

    def reset_classifier(self, dim_out, global_pool=''):
        self.dim_out = dim_out
        self.head = nn.Linear(self.dim_feat, dim_out) if dim_out > 0 else nn.Identity()
        self._param_names = [n for n, _ in self.named_parameters()]

    # ------------------------------------------------------------------ checks
    def _check(self, x):
        if x.dim() != 4 or x.shape[2] != self.num_joints or x.shape[3] != self.dim_in:
            raise ValueError(f'expected input [B, T, {self.num_joints}, {self.dim_in}], got {tuple(x.shape)}')
        if not 1 <= x.shape[1] <= self.maxlen:
            raise ValueError(f'sequence length {x.shape[1]} outside [1, maxlen={self.maxlen}]')
        if not x.is_cuda:
            raise RuntimeError('motionbert_amd.DSTformer has no CPU path: move the model and input to a ROCm device '
                               '(the arithmetic lives in libmbx.so, gfx950 HIP kernels)')
        if not self.dim_rep:
            raise NotImplementedError('dim_rep=0 (no pre_logits layer) is not used by any shipped config')
        if self.precision not in _DTYPES:
            raise ValueError(f"precision must be one of {list(_DTYPES)}, got {self.precision!r}")
        if isinstance(self.head, nn.Linear) and self.head.in_features != self.dim_rep:
            raise NotImplementedError('reset_classifier() head with in_features != dim_rep cannot follow pre_logits')

    def forward(self, x, return_rep=False):
        self._check(x)
        if x.shape[0] == 0:   # empty batch: same shapes as the reference (reshape(-1, J, C) of nothing), zero gradients
            return_rep = return_rep or not isinstance(self.head, nn.Linear)
            out = x.new_zeros((0, x.shape[1], self.num_joints, self.dim_rep if return_rep else self.dim_out), dtype=torch.float32)
            if not torch.is_grad_enabled():
                return out
            names, params = named_parameter_tensors(self)
            return out + sum(p.sum() for n, p in zip(names, params) if not (return_rep and n.startswith('head.'))) * 0.0
        from . import hip_ops
        x = x.contiguous().float()
        # dim_out <= 0 / reset_classifier(0): the head is nn.Identity (DSTformer.py:300,326) -> forward returns the representation
        if return_rep == 'flip_tta':   # motionbert_amd.augment.flip_tta: evaluation only (pose output, no gradient)
            if torch.is_grad_enabled() or not isinstance(self.head, nn.Linear):
                raise RuntimeError('flip test-time augmentation is an evaluation path: call it under torch.no_grad() on the pose head')
            from .augment import FLIP_PERM
            perm = torch.tensor(FLIP_PERM, dtype=torch.int32, device=x.device)
            return run(hip_ops.get(), self, x, ('tta', perm), None)
        return run(hip_ops.get(), self, x, return_rep or not isinstance(self.head, nn.Linear), getattr(self, '_grad_sync', None))

    def get_representation(self, x):
        return self.forward(x, return_rep=True)

    def get_pooled_representation(self, x, persons: int = 1, dropout: float = 0.0):
        """`[B, T, J, dim_in] -> [B / persons, J, dim_rep]`: the representation averaged over the T frames and over `persons`
        consecutive clips, with element-wise dropout before the means in training -- exactly what the reference's action
        heads do first with `get_representation(x)` (lib/model/model_action.py:15-24,37-45: dropout, mean over T, mean over
        M), fused onto the backbone tail so that neither the [B,T,J,dim_rep] cotangent nor the dropout mask ever exists in
        memory.  Used by motionbert_amd.action.ActionNet."""
        p = float(dropout) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0      # from torch's global CPU generator
        return self.forward(x, return_rep=('pool', int(persons), p, seed))
