"""Diagnostics: run one kernel ("victim") repeatedly on fixed inputs on stream A while another kernel ("aggressor")
loops on stream B, and count how many victim launches produce output that differs bit-wise from the first one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionbert_amd import hip_ops
from motionbert_amd.engine import EPI_STORE, EPI_DGELU, EPI_RESID, EPI_GELU, MODE_SPATIAL, MODE_TEMPORAL
ops = hip_ops.get()
dev = 'cuda'
B, T, J, C, H, HID = 8, 243, 17, 512, 8, 1024
M = B * T * J
torch.manual_seed(0)
f = lambda *s: torch.randn(*s, device=dev) * 0.5
t = lambda *s: f(*s).bfloat16()
bf = torch.bfloat16

def mk(kind):
    """returns (launch(), outputs list)"""
    if kind == 'ln_bwd':
        dy, x, g, dres = t(M, C), f(M, C), f(C), f(M, C)
        mean, rstd = x.mean(1), 1 / (x.var(1, unbiased=False) + 1e-6).sqrt()
        dx, dg, db = f(M, C), f(C), f(C)
        return (lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres, None, dx, None, dg, db)), [dx, dg, db]
    if kind == 'ln_fwd':
        x, g, b = f(M, C), f(C), f(C)
        y, mean, rstd = t(M, C), f(M), f(M)
        return (lambda: ops.layernorm_fwd(x, g, b, 1e-6, y, mean, rstd)), [y, mean, rstd]
    if kind.startswith('nt'):
        N, K = {'nt_qkv': (3 * C, C), 'nt_fc1': (HID, C), 'nt_fc2': (C, HID), 'nt_dgelu': (HID, C)}[kind]
        a, w, bias = t(M, K), t(N, K), f(N)
        if kind == 'nt_dgelu':
            out, aux = t(M, N), t(M, N)
            return (lambda: ops.gemm_nt(a, w, None, EPI_DGELU, out_t=out, aux_t=aux)), [out]
        if kind == 'nt_fc2':
            out, res = f(M, N), f(M, N)
            return (lambda: ops.gemm_nt(a, w, bias, EPI_RESID, resid=res, out_f=out)), [out]
        out = t(M, N)
        return (lambda: ops.gemm_nt(a, w, bias, EPI_STORE, out_t=out)), [out]
    if kind.startswith('tn'):
        N, K = {'tn_qkv': (3 * C, C), 'tn_fc1': (HID, C), 'tn_proj': (C, C)}[kind]
        dy, a, dw, db = t(M, N), t(M, K), f(N, K), f(N)
        return (lambda: ops.gemm_tn(dy, a, dw, db)), [dw, db]
    if kind.startswith('attn'):
        mode = MODE_SPATIAL if kind.endswith('_s') else MODE_TEMPORAL
        qkv, o, lse = t(M, 3 * C), t(M, C), f(M, H)
        ops.attn_fwd(qkv, o, lse, B, T, J, H, 0.125, mode)
        if kind.startswith('attn_fwd'):
            return (lambda: ops.attn_fwd(qkv, o, lse, B, T, J, H, 0.125, mode)), [o, lse]
        do, dqkv = t(M, C), t(M, 3 * C)
        return (lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, T, J, H, 0.125, mode)), [dqkv]
    if kind == 'fuse_bwd':
        dh, xs, xt, w = f(M, C), f(M, C), f(M, C), f(2, 2 * C)
        alpha = torch.softmax(f(M, 2), -1)
        outs = [f(M, C), f(M, C), t(M, C), t(M, C), f(2, 2 * C), f(2)]
        return (lambda: ops.fuse_bwd(dh, xs, xt, alpha, w, *outs)), outs
    raise KeyError(kind)

KINDS = ['ln_bwd', 'ln_fwd', 'nt_qkv', 'nt_fc2', 'nt_dgelu', 'tn_qkv', 'tn_proj', 'attn_fwd_s', 'attn_fwd_t', 'attn_bwd_s', 'attn_bwd_t', 'fuse_bwd']
victims = sys.argv[1].split(',') if len(sys.argv) > 1 else KINDS
aggressors = sys.argv[2].split(',') if len(sys.argv) > 2 else ['none'] + KINDS
ITER = int(os.environ.get('ITER', 40))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for v in victims:
    vl, vo = mk(v)
    vl(); torch.cuda.synchronize()
    ref = [o.clone() for o in vo]
    row = []
    for a in aggressors:
        al = mk(a)[0] if a != 'none' else None
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for it in range(ITER):
            if al is not None:
                with torch.cuda.stream(sb):
                    al(); al()
            with torch.cuda.stream(sa):
                vl()
                ne = sum((o.view(torch.int16 if o.element_size() == 2 else torch.int32) != r.view(torch.int16 if r.element_size() == 2 else torch.int32)).any().to(torch.int64) for o, r in zip(vo, ref))
                bad += (ne > 0).to(torch.int64)
        torch.cuda.synchronize()
        row.append(f'{a}:{int(bad)}')
    print(f'victim {v:11s} mismatching launches of {ITER}:  ' + '  '.join(row), flush=True)
