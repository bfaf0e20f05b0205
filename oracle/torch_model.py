"""TEST / MEASUREMENT INFRASTRUCTURE ONLY: the DSTformer forward as ONE plain-PyTorch function over a reference-format state_dict.

Why it exists: bench.py's `cpu_baseline` leg has to time "the reference PyTorch model on the host cores" on a GPU box that has no
reference checkout.  The round-3 port drove the product's kernel-by-kernel sequencing through oracle/torch_ops.MockOps (unfused
copies, hand-written backward) and ran at 0.4-0.7x of the real reference -- it understated the baseline.  This restatement issues
the SAME ATen operator mix the reference does (addmm / bmm / softmax / native_layer_norm / gelu, autograd backward) from the same
parameters, so its speed is the reference's (tools/cpu_calibration.py: ratio and spread in profiles/r04_cpu_calibration.txt).
Follows lib/model/DSTformer.py: MLP.forward :79-85, Attention.forward :139-150 with forward_spatial :178-186 and forward_temporal
:188-200, Block.forward :239-249 (st_mode stage_st / stage_ts), DSTformer.forward :329-358.  Written against the state_dict keys
(SURVEY.md 8b), not against the module classes; dropout / drop-path are identities at the rates every shipped config uses.
Like everything under oracle/ it is never imported by the package."""
import torch
import torch.nn.functional as F


def _attention(P, pre, x, heads, mode, frames):
    """x [B*T, J, C] -> [B*T, J, C]; mode 's': softmax over the J joints of a frame, 't': over the T frames of a joint."""
    BT, J, C = x.shape
    hd = C // heads
    qkv = F.linear(x, P[pre + '.qkv.weight'], P.get(pre + '.qkv.bias')).view(BT, J, 3, heads, hd)
    q, k, v = qkv.unbind(2)                                   # each [BT, J, H, hd]
    if mode == 's':
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))      # [BT, H, J, hd]
        a = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, -1)
        o = (a @ v).transpose(1, 2).reshape(BT, J, C)
    else:
        B = BT // frames
        q, k, v = (t.reshape(B, frames, J, heads, hd).permute(0, 3, 2, 1, 4) for t in (q, k, v))   # [B, H, J, T, hd]
        a = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, -1)
        o = (a @ v).permute(0, 3, 2, 1, 4).reshape(BT, J, C)
    return F.linear(o, P[pre + '.proj.weight'], P[pre + '.proj.bias'])


def _mlp(P, pre, x):
    return F.linear(F.gelu(F.linear(x, P[pre + '.fc1.weight'], P[pre + '.fc1.bias'])), P[pre + '.fc2.weight'], P[pre + '.fc2.bias'])


def _ln(P, pre, x, eps):
    return F.layer_norm(x, x.shape[-1:], P[pre + '.weight'], P[pre + '.bias'], eps)


def _block(P, pre, x, heads, frames, eps, order):
    for sfx in order:                                         # ('s', 't') for blocks_st, ('t', 's') for blocks_ts
        x = x + _attention(P, f'{pre}.attn_{sfx}', _ln(P, f'{pre}.norm1_{sfx}', x, eps), heads, sfx, frames)
        x = x + _mlp(P, f'{pre}.mlp_{sfx}', _ln(P, f'{pre}.norm2_{sfx}', x, eps))
    return x


def forward(P, x, depth, heads, eps=1e-6, return_rep=False):
    """P: {state_dict key: tensor}; x [B, T, J, dim_in] -> [B, T, J, dim_out] (or the representation)."""
    B, T, J, _ = x.shape
    h = F.linear(x.reshape(B * T, J, -1), P['joints_embed.weight'], P['joints_embed.bias']) + P['pos_embed']
    C = h.shape[-1]
    h = (h.view(B, T, J, C) + P['temp_embed'][:, :T]).view(B * T, J, C)
    for i in range(depth):
        a = _block(P, f'blocks_st.{i}', h, heads, T, eps, ('s', 't'))
        b = _block(P, f'blocks_ts.{i}', h, heads, T, eps, ('t', 's'))
        if f'ts_attn.{i}.weight' in P:
            w = torch.softmax(F.linear(torch.cat([a, b], -1), P[f'ts_attn.{i}.weight'], P[f'ts_attn.{i}.bias']), -1)
            h = a * w[..., 0:1] + b * w[..., 1:2]
        else:
            h = (a + b) * 0.5
    h = torch.tanh(F.linear(_ln(P, 'norm', h, eps).view(B, T, J, C), P['pre_logits.fc.weight'], P['pre_logits.fc.bias']))
    return h if return_rep else F.linear(h, P['head.weight'], P['head.bias'])
