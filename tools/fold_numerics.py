"""Numerics of the two backward formulations in bf16 (CPU, torch restatement of the kernel set = oracle/torch_ops.py MockOps) on
the reference-minted fixture full_1x243 (real fp64 gradients of the reference):
    python tools/fold_numerics.py            # plain / folded / folded with the LayerNorm backward done exactly from the accumulator
    python tools/fold_numerics.py seeds      # four rounding realisations (weights perturbed by 1e-6) of plain and folded
Output of both runs: profiles/r03_fold_numerics.txt."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import build_model, load_golden, trained_like, rel_l2
from tests.test_gpu_model import _fixture_grad_errors
from oracle.torch_ops import MockOps
from motionbert_amd import model as M
torch.set_num_threads(8)
name = 'full_1x243'
if len(sys.argv) > 1 and sys.argv[1] == 'seeds':
    z, cfg = load_golden(name)

    keys = ['blocks_ts.0.attn_t.qkv.weight', 'blocks_ts.0.mlp_s.fc2.weight', 'ts_attn.0.weight', 'blocks_st.0.attn_s.qkv.weight', 'blocks_st.3.mlp_s.fc1.weight']
    for pseed in (0, 1, 2, 3):
        for tag, fold in (('plain', False), ('fold', True)):
            model = build_model(cfg, seed=0)
            trained_like(model, int(z['trained_seed']))
            if pseed:
                g = torch.Generator().manual_seed(100 + pseed)
                with torch.no_grad():
                    for p in model.parameters():
                        p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=g))
            model.precision, model.fold_ln = 'bf16', fold
            x = torch.from_numpy(z['x']).requires_grad_(True)
            out = M.run(MockOps(), model, x)
            (out * torch.from_numpy(z['cot'])).sum().backward()
            e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
            print(pseed, tag, f'out {rel_l2(out.detach().numpy(), z["out"]):.4f} glob {e_all:.4f}', ' '.join(f'{per[k]:.4f}' for k in keys), flush=True)
    sys.exit(0)

z, cfg = load_golden(name)
res = {}
for tag, fold, patch in (('plain', False, None), ('fold', True, None), ('fold_exactY', True, 'exactY')):
    model = build_model(cfg, seed=0)
    if int(z['trained_seed']) >= 0:
        trained_like(model, int(z['trained_seed']))
    model.precision, model.fold_ln = 'bf16', fold
    ops = MockOps()
    if patch == 'exactY':
        # c2 from the UNROUNDED forward output: recompute Y0 = xhat W'^T in fp32 inside the stats ops
        import types
        def lnbwd(self, a_t, w_t, xhat, rowc, dres, extra, dx, dx_t):
            acc = a_t.float() @ w_t.float().t()
            xh = xhat.float()
            rs = rowc[:, 0:1]
            r = dres + rs * (acc - acc.mean(-1, keepdim=True) - xh * (acc * xh).mean(-1, keepdim=True))
            if extra is not None: r = r + extra
            dx.copy_(r)
            if dx_t is not None: dx_t.copy_(r.to(dx_t.dtype))
        ops.gemm_nt_lnbwd = types.MethodType(lnbwd, ops)
    x = torch.from_numpy(z['x']).requires_grad_(True)
    t0 = time.time()
    out = M.run(ops, model, x)
    (out * torch.from_numpy(z['cot'])).sum().backward()
    e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
    res[tag] = per
    print(tag, 'out', rel_l2(out.detach().numpy(), z['out']), 'dx', rel_l2(x.grad.numpy(), z['dx']), 'grad_global', e_all, 'worst', worst, e_worst, f'{time.time()-t0:.0f}s', flush=True)
names = [str(n) for n in z['names']]
ac = dict(zip(names, z['autocast_grad_per']))
top = sorted(res['fold'], key=lambda n: -res['fold'][n])[:14]
for n in top:
    print(f'{n:40s} plain {res["plain"][n]:.4f} fold {res["fold"][n]:.4f} fold_exactY {res["fold_exactY"][n]:.4f} ref_autocast {ac[n]:.4f}')
