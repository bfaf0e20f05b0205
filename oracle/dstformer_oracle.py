"""CPU oracle for the DSTformer hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy float64 *restatement* of the algorithm implemented by the
reference model `lib/model/DSTformer.py` (Walter0807/MotionBERT).  It is the
checker that the HIP kernels are compared against.  It must never be imported
by product code: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may use it.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 8c), so the oracle is pinned against the reference model
itself, imported read-only in the build container by
`oracle/make_golden.py`, which writes the fixtures under `tests/golden/`.
`tests/test_oracle_golden.py` re-checks the oracle against those fixtures on
every run (no access to /root/reference needed).

Every function cites the reference lines it follows (paths relative to the
reference checkout).  The arithmetic that the reference delegates to
PyTorch/ATen is restated from its published semantics:
  Linear     y = x W^T + b
  LayerNorm  biased variance, eps inside the sqrt
  GELU       exact erf form  0.5 x (1 + erf(x / sqrt 2))
  softmax    max-subtracted
The backward pass (implicit autograd in the reference, `train.py:205`) is
written out by hand here and checked against the reference's autograd.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
from scipy.special import erf as _erf

F64 = np.float64


@dataclass
class OracleConfig:
    """Constructor arguments of DSTformer that change the arithmetic
    (DSTformer.py:270-273; eps comes from lib/utils/learning.py:84)."""
    dim_in: int = 3
    dim_out: int = 3
    dim_feat: int = 256
    dim_rep: int = 512
    depth: int = 5
    num_heads: int = 8
    mlp_ratio: float = 4
    num_joints: int = 17
    maxlen: int = 243
    eps: float = 1e-5
    qk_scale: float | None = None
    att_fuse: bool = True

    @property
    def hidden(self) -> int:
        return int(self.dim_feat * self.mlp_ratio)  # DSTformer.py:232

    @property
    def scale(self) -> float:
        return self.qk_scale or (self.dim_feat // self.num_heads) ** -0.5  # DSTformer.py:94


# --------------------------------------------------------------------------
# elementary ops (each usable on its own by the per-kernel parity tests)
# --------------------------------------------------------------------------

def linear_fwd(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


def linear_bwd(dy, x, w):
    """dx, dW, db of y = x W^T + b (x, dy flattened to 2-D)."""
    x2 = x.reshape(-1, x.shape[-1])
    dy2 = dy.reshape(-1, dy.shape[-1])
    return (dy2 @ w).reshape(x.shape), dy2.T @ x2, dy2.sum(0)


def layernorm_fwd(x, g, b, eps):
    """nn.LayerNorm over the last dim (DSTformer.py:221-222,230-231,292)."""
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * g + b, xhat, rstd


def layernorm_bwd(dy, xhat, rstd, g):
    dg = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx, dg, db


def gelu_fwd(u):
    """nn.GELU() default = exact erf form (DSTformer.py:70,75)."""
    return 0.5 * u * (1.0 + _erf(u / math.sqrt(2.0)))


def gelu_grad(u):
    return 0.5 * (1.0 + _erf(u / math.sqrt(2.0))) + u * np.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)


def softmax(s):
    e = np.exp(s - s.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def attention_fwd(qkv, B, T, J, H, scale, mode):
    """Multi-head attention core on a packed qkv tensor.

    qkv: [B*T, J, 3*C] with channel order [3][H][hd] (DSTformer.py:143).
    mode 'spatial'  = forward_spatial  (DSTformer.py:178-186): softmax over the J joints of a frame.
    mode 'temporal' = forward_temporal (DSTformer.py:188-200): softmax over the T frames of a joint.
    Returns o [B*T, J, C] (heads concatenated head-major) and the probabilities."""
    C3 = qkv.shape[-1]
    C = C3 // 3
    hd = C // H
    q5 = qkv.reshape(B, T, J, 3, H, hd)
    q, k, v = q5[:, :, :, 0], q5[:, :, :, 1], q5[:, :, :, 2]  # [B,T,J,H,hd]
    if mode == 'spatial':
        s = np.einsum('btihd,btjhd->bthij', q, k) * scale
        p = softmax(s)
        o = np.einsum('bthij,btjhd->btihd', p, v)
    elif mode == 'temporal':
        s = np.einsum('bsjhd,btjhd->bjhst', q, k) * scale
        p = softmax(s)
        o = np.einsum('bjhst,btjhd->bsjhd', p, v)
    else:
        raise ValueError(mode)
    return o.reshape(B * T, J, C), p


def attention_bwd(do, qkv, p, B, T, J, H, scale, mode):
    """Gradient of attention_fwd w.r.t. the packed qkv tensor."""
    C = qkv.shape[-1] // 3
    hd = C // H
    q5 = qkv.reshape(B, T, J, 3, H, hd)
    q, k, v = q5[:, :, :, 0], q5[:, :, :, 1], q5[:, :, :, 2]
    do5 = do.reshape(B, T, J, H, hd)
    dq5 = np.zeros_like(q5)
    if mode == 'spatial':
        dp = np.einsum('btihd,btjhd->bthij', do5, v)
        ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * scale
        dq5[:, :, :, 0] = np.einsum('bthij,btjhd->btihd', ds, k)
        dq5[:, :, :, 1] = np.einsum('bthij,btihd->btjhd', ds, q)
        dq5[:, :, :, 2] = np.einsum('bthij,btihd->btjhd', p, do5)
    else:
        dp = np.einsum('bsjhd,btjhd->bjhst', do5, v)
        ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * scale
        dq5[:, :, :, 0] = np.einsum('bjhst,btjhd->bsjhd', ds, k)
        dq5[:, :, :, 1] = np.einsum('bjhst,bsjhd->btjhd', ds, q)
        dq5[:, :, :, 2] = np.einsum('bjhst,bsjhd->btjhd', p, do5)
    return dq5.reshape(qkv.shape)


def embed_fwd(x, w, b, pos, temp):
    """joints_embed + pos_embed + temp_embed (DSTformer.py:330-337).
    x [B,T,J,3] -> [B,T,J,C]; pos [1,J,C]; temp [1,maxlen,1,C]."""
    B, T, J, _ = x.shape
    y = x @ w.T + b
    return y + pos.reshape(1, 1, J, -1) + temp[:, :T]


def embed_bwd(dy, x, w, maxlen):
    B, T, J, C = dy.shape
    dw = dy.reshape(-1, C).T @ x.reshape(-1, x.shape[-1])
    db = dy.reshape(-1, C).sum(0)
    dpos = dy.sum((0, 1)).reshape(1, J, C)
    dtemp = np.zeros((1, maxlen, 1, C), dy.dtype)
    dtemp[0, :T, 0] = dy.sum((0, 2))
    dx = dy @ w
    return dx, dw, db, dpos, dtemp


def fuse_fwd(x_st, x_ts, w, b):
    """Adaptive fusion (DSTformer.py:343-349): alpha = softmax(Linear(2C,2)(cat[x_st,x_ts]))."""
    logits = np.concatenate([x_st, x_ts], -1) @ w.T + b
    alpha = softmax(logits)
    return x_st * alpha[..., 0:1] + x_ts * alpha[..., 1:2], alpha


def fuse_bwd(dx, x_st, x_ts, alpha, w):
    C = x_st.shape[-1]
    da = np.stack([(dx * x_st).sum(-1), (dx * x_ts).sum(-1)], -1)
    dl = alpha * (da - (da * alpha).sum(-1, keepdims=True))
    cat = np.concatenate([x_st, x_ts], -1)
    dw = dl.reshape(-1, 2).T @ cat.reshape(-1, 2 * C)
    db = dl.reshape(-1, 2).sum(0)
    dcat = dl @ w
    dx_st = dx * alpha[..., 0:1] + dcat[..., :C]
    dx_ts = dx * alpha[..., 1:2] + dcat[..., C:]
    return dx_st, dx_ts, dw, db


# --------------------------------------------------------------------------
# sub-layers and blocks
# --------------------------------------------------------------------------

def _attn_sublayer_fwd(x, P, pre, norm, attn, cfg, B, T, mode):
    """x + proj(attention(qkv(norm(x))))   (DSTformer.py:241,243 / 139-149)."""
    J = cfg.num_joints
    xn, xhat, rstd = layernorm_fwd(x, P[f'{pre}.{norm}.weight'], P[f'{pre}.{norm}.bias'], cfg.eps)
    qkv = linear_fwd(xn, P[f'{pre}.{attn}.qkv.weight'], P.get(f'{pre}.{attn}.qkv.bias'))
    o, p = attention_fwd(qkv, B, T, J, cfg.num_heads, cfg.scale, mode)
    y = x + linear_fwd(o, P[f'{pre}.{attn}.proj.weight'], P[f'{pre}.{attn}.proj.bias'])
    return y, dict(xn=xn, xhat=xhat, rstd=rstd, qkv=qkv, o=o, p=p)


def _attn_sublayer_bwd(dy, c, P, G, pre, norm, attn, cfg, B, T, mode):
    J = cfg.num_joints
    do, G[f'{pre}.{attn}.proj.weight'], G[f'{pre}.{attn}.proj.bias'] = linear_bwd(dy, c['o'], P[f'{pre}.{attn}.proj.weight'])
    dqkv = attention_bwd(do, c['qkv'], c['p'], B, T, J, cfg.num_heads, cfg.scale, mode)
    dxn, G[f'{pre}.{attn}.qkv.weight'], dbq = linear_bwd(dqkv, c['xn'], P[f'{pre}.{attn}.qkv.weight'])
    if f'{pre}.{attn}.qkv.bias' in P:
        G[f'{pre}.{attn}.qkv.bias'] = dbq
    dx, G[f'{pre}.{norm}.weight'], G[f'{pre}.{norm}.bias'] = layernorm_bwd(dxn, c['xhat'], c['rstd'], P[f'{pre}.{norm}.weight'])
    return dy + dx


def _mlp_sublayer_fwd(x, P, pre, norm, mlp, cfg):
    """x + fc2(gelu(fc1(norm(x))))   (DSTformer.py:242,244 / 79-85)."""
    xn, xhat, rstd = layernorm_fwd(x, P[f'{pre}.{norm}.weight'], P[f'{pre}.{norm}.bias'], cfg.eps)
    u = linear_fwd(xn, P[f'{pre}.{mlp}.fc1.weight'], P[f'{pre}.{mlp}.fc1.bias'])
    g = gelu_fwd(u)
    y = x + linear_fwd(g, P[f'{pre}.{mlp}.fc2.weight'], P[f'{pre}.{mlp}.fc2.bias'])
    return y, dict(xn=xn, xhat=xhat, rstd=rstd, u=u, g=g)


def _mlp_sublayer_bwd(dy, c, P, G, pre, norm, mlp, cfg):
    dg, G[f'{pre}.{mlp}.fc2.weight'], G[f'{pre}.{mlp}.fc2.bias'] = linear_bwd(dy, c['g'], P[f'{pre}.{mlp}.fc2.weight'])
    du = dg * gelu_grad(c['u'])
    dxn, G[f'{pre}.{mlp}.fc1.weight'], G[f'{pre}.{mlp}.fc1.bias'] = linear_bwd(du, c['xn'], P[f'{pre}.{mlp}.fc1.weight'])
    dx, G[f'{pre}.{norm}.weight'], G[f'{pre}.{norm}.bias'] = layernorm_bwd(dxn, c['xhat'], c['rstd'], P[f'{pre}.{norm}.weight'])
    return dy + dx


# sub-layer order of the two Block flavours (DSTformer.py:240-249)
_ORDER = {
    'st': [('attn', 'norm1_s', 'attn_s', 'spatial'), ('mlp', 'norm2_s', 'mlp_s', None),
           ('attn', 'norm1_t', 'attn_t', 'temporal'), ('mlp', 'norm2_t', 'mlp_t', None)],
    'ts': [('attn', 'norm1_t', 'attn_t', 'temporal'), ('mlp', 'norm2_t', 'mlp_t', None),
           ('attn', 'norm1_s', 'attn_s', 'spatial'), ('mlp', 'norm2_s', 'mlp_s', None)],
}


def block_fwd(x, P, pre, kind, cfg, B, T):
    caches = []
    for typ, norm, mod, mode in _ORDER[kind]:
        if typ == 'attn':
            x, c = _attn_sublayer_fwd(x, P, pre, norm, mod, cfg, B, T, mode)
        else:
            x, c = _mlp_sublayer_fwd(x, P, pre, norm, mod, cfg)
        caches.append(c)
    return x, caches


def block_bwd(dy, caches, P, G, pre, kind, cfg, B, T):
    for (typ, norm, mod, mode), c in zip(reversed(_ORDER[kind]), reversed(caches)):
        if typ == 'attn':
            dy = _attn_sublayer_bwd(dy, c, P, G, pre, norm, mod, cfg, B, T, mode)
        else:
            dy = _mlp_sublayer_bwd(dy, c, P, G, pre, norm, mod, cfg)
    return dy


# --------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------

def forward(P, x, cfg: OracleConfig, return_rep=False, want_cache=False):
    """DSTformer.forward (DSTformer.py:329-358).  P: dict name -> float64 array
    keyed like the reference state_dict.  x: [B,T,J,dim_in]."""
    P = {k: np.asarray(v, F64) for k, v in P.items()}
    x = np.asarray(x, F64)
    B, T, J, _ = x.shape
    C = cfg.dim_feat
    h = embed_fwd(x, P['joints_embed.weight'], P['joints_embed.bias'], P['pos_embed'], P['temp_embed'])
    h = h.reshape(B * T, J, C)
    cache = dict(x=x, levels=[])
    for i in range(cfg.depth):
        x_st, c_st = block_fwd(h, P, f'blocks_st.{i}', 'st', cfg, B, T)
        x_ts, c_ts = block_fwd(h, P, f'blocks_ts.{i}', 'ts', cfg, B, T)
        if cfg.att_fuse:
            h, alpha = fuse_fwd(x_st, x_ts, P[f'ts_attn.{i}.weight'], P[f'ts_attn.{i}.bias'])
        else:
            h, alpha = (x_st + x_ts) * 0.5, None  # DSTformer.py:351
        cache['levels'].append(dict(st=c_st, ts=c_ts, x_st=x_st, x_ts=x_ts, alpha=alpha))
    xn, xhat, rstd = layernorm_fwd(h, P['norm.weight'], P['norm.bias'], cfg.eps)
    cache.update(xn=xn, xhat=xhat, rstd=rstd)
    if cfg.dim_rep:
        rep = np.tanh(linear_fwd(xn, P['pre_logits.fc.weight'], P['pre_logits.fc.bias']))
    else:
        rep = xn
    cache['rep'] = rep
    rep4 = rep.reshape(B, T, J, -1)
    if return_rep:
        out = rep4
    elif cfg.dim_out > 0:
        out = linear_fwd(rep4, P['head.weight'], P['head.bias'])
    else:
        out = rep4
    return (out, cache) if want_cache else out


def backward(P, cache, dout, cfg: OracleConfig, return_rep=False):
    """Gradients of `forward` w.r.t. every parameter and the input, given d(out)."""
    P = {k: np.asarray(v, F64) for k, v in P.items()}
    dout = np.asarray(dout, F64)
    x = cache['x']
    B, T, J, _ = x.shape
    C = cfg.dim_feat
    G = {}
    if return_rep or cfg.dim_out <= 0:
        drep = dout.reshape(B * T, J, -1)
        if cfg.dim_out > 0:
            G['head.weight'] = np.zeros_like(P['head.weight'])
            G['head.bias'] = np.zeros_like(P['head.bias'])
    else:
        drep, G['head.weight'], G['head.bias'] = linear_bwd(dout.reshape(B * T, J, -1), cache['rep'], P['head.weight'])
    if cfg.dim_rep:
        dpre = drep * (1.0 - cache['rep'] ** 2)
        dxn, G['pre_logits.fc.weight'], G['pre_logits.fc.bias'] = linear_bwd(dpre, cache['xn'], P['pre_logits.fc.weight'])
    else:
        dxn = drep
    dh, G['norm.weight'], G['norm.bias'] = layernorm_bwd(dxn, cache['xhat'], cache['rstd'], P['norm.weight'])
    for i in reversed(range(cfg.depth)):
        lv = cache['levels'][i]
        if cfg.att_fuse:
            d_st, d_ts, G[f'ts_attn.{i}.weight'], G[f'ts_attn.{i}.bias'] = fuse_bwd(
                dh, lv['x_st'], lv['x_ts'], lv['alpha'], P[f'ts_attn.{i}.weight'])
        else:
            d_st = d_ts = dh * 0.5
        d1 = block_bwd(d_st, lv['st'], P, G, f'blocks_st.{i}', 'st', cfg, B, T)
        d2 = block_bwd(d_ts, lv['ts'], P, G, f'blocks_ts.{i}', 'ts', cfg, B, T)
        dh = d1 + d2
    dx, G['joints_embed.weight'], G['joints_embed.bias'], G['pos_embed'], G['temp_embed'] = embed_bwd(
        dh.reshape(B, T, J, C), x, P['joints_embed.weight'], cfg.maxlen)
    return G, dx


# --------------------------------------------------------------------------
# training-loss restatement used by the golden fixtures (config 3 of
# BASELINE.json: loss_mpjpe + 0.5 n_mpjpe + 20 loss_velocity,
# configs/pose3d/MB_train_h36m.yaml:38-39; lib/model/loss.py:56-66,79-96,133-142)
# is deliberately NOT part of the oracle: the golden gradients use
# d(out) = a fixed random cotangent instead, so only the hot path is pinned.
# --------------------------------------------------------------------------

def rel_l2(a, b):
    a = np.asarray(a, F64)
    b = np.asarray(b, F64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
