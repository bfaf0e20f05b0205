"""The raw-operand LayerNorm carried into TRAINING (VERDICT r3 missing item 3; DESIGN "Next" 0a), as a CPU emulation against the frozen
gates before any kernel is written.  With the torch restatement of the kernel set (oracle/torch_ops.MockOps, bf16, folded sequencing):
  forward   Linear'(LayerNorm(y)) = rstd (bf16(y) . W'^T - mean rsum) + b'          (the operand is the RAW row; no xhat is written)
  backward  dW' = (rstd o dY)^T . bf16(y) - v 1^T,  v[n] = sum_r mean_r (rstd o dY)[r, n]     (dY leaves its producer scaled by rstd)
            db' = sum_r (rstd o dY)[r, n] / rstd_r
            dx  = dres + (rstd o dY) . W' - rstd c1 - xhat rstd c2,   xhat = (bf16(y) - mean) rstd  rebuilt in the epilogue
on the reference-minted fixtures, four rounding realisations, against the gates of tests/test_gpu_model.py.  The scaled dY is emulated
as bf16(rstd * float(bf16(dY))) -- one rounding more than a producer that scales before it rounds: pessimistic.
    python tools/rawtrain_numerics.py [fixture]      -> profiles/r04_rawtrain_numerics.txt"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import model as M                      # noqa: E402
from oracle.torch_ops import MockOps, EPI_GELU, EPI_STORE  # noqa: E402
from tests.helpers import build_model, load_golden, rel_l2, trained_like   # noqa: E402
from tests.test_gpu_model import _fixture_grad_errors      # noqa: E402

torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else 'full_1x243'
z, cfg = load_golden(name)
names = [str(n) for n in z['names']]
ac_per = dict(zip(names, (float(a) for a in z['autocast_grad_per'])))
ac_out, ac_glob = float(z['autocast_out']), float(z['autocast_grad_global'])
BF = torch.bfloat16


def raw_training(ops):
    raw = []                           # the xhat tensors the engine passes around carry their raw rows as an attribute: (bf16(y), mean, rstd)
    p_ln, p_nt, p_tn, p_lnb = MockOps.layernorm_fwd, MockOps.gemm_nt, MockOps.gemm_tn, MockOps.gemm_nt_lnbwd

    def layernorm_fwd(self, x, g, b, eps, y_t, mean, rstd):
        p_ln(self, x, g, b, eps, y_t, mean, rstd)
        if g is None and not isinstance(y_t, tuple):
            y_t._raw = (x.to(BF), mean.clone(), rstd.clone())
            raw.append(1)

    def gemm_nt(self, a_t, w_t, bias, epi, out_t=None, out2_t=None, out_f=None, resid=None, aux_t=None):
        r = getattr(a_t, '_raw', None)
        if r is None or epi not in (EPI_STORE, EPI_GELU):
            return p_nt(self, a_t, w_t, bias, epi, out_t, out2_t, out_f, resid, aux_t)
        self._log(f'gemm_nt.{epi}')
        y, mean, rstd = r
        acc = rstd[:, None] * (y.float() @ w_t.float().t() - mean[:, None] * w_t.float().sum(1)) + bias
        if epi == EPI_STORE:
            out_t.copy_(acc.to(out_t.dtype))
        else:
            if out_t is not None:
                out_t.copy_(acc.to(out_t.dtype))
            out2_t.copy_(torch.nn.functional.gelu(acc).to(out2_t.dtype))

    def gemm_tn(self, dy_t, a_t, dw, db):
        r = getattr(a_t, '_raw', None)
        if r is None:
            return p_tn(self, dy_t, a_t, dw, db)
        self._log('gemm_tn')
        y, mean, rstd = r
        dys = (rstd[:, None] * dy_t.float()).to(BF).float()
        dw.copy_(dys.t() @ y.float() - (mean[:, None] * dys).sum(0)[:, None])
        if db is not None:
            db.copy_((dys / rstd[:, None]).sum(0))

    def gemm_nt_lnbwd(self, a_t, w_t, xhat, rowc, dres, extra, dx, dx_t):
        r = getattr(xhat, '_raw', None)
        if r is None:
            return p_lnb(self, a_t, w_t, xhat, rowc, dres, extra, dx, dx_t)
        self._log('gemm_nt.lnbwd' + ('' if dx is not None else '.stream'))
        y, mean, rstd = r
        dys = (rstd[:, None] * a_t.float()).to(BF).float()
        xh = (y.float() - mean[:, None]) * rstd[:, None]
        res = dres.float() + dys @ w_t.float().t() - rowc[:, 1:2] - xh * rowc[:, 2:3]
        if extra is not None:
            res = res + extra
        if dx is not None:
            dx.copy_(res)
        if dx_t is not None:
            dx_t.copy_(res.to(dx_t.dtype))

    for n, f in (('layernorm_fwd', layernorm_fwd), ('gemm_nt', gemm_nt), ('gemm_tn', gemm_tn), ('gemm_nt_lnbwd', gemm_nt_lnbwd)):
        setattr(ops, n, types.MethodType(f, ops))
    return ops, raw


for pseed in (0, 1, 2, 3):
    for tag in ('xhat operand (product)', 'raw operand, fwd + bwd'):
        model = build_model(cfg, seed=0)
        if int(z['trained_seed']) >= 0:
            trained_like(model, int(z['trained_seed']))
        if pseed:
            g = torch.Generator().manual_seed(100 + pseed)
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=g))
        model.precision = 'bf16'
        ops, nraw = MockOps(), None
        if tag.startswith('raw'):
            ops, nraw = raw_training(ops)
        x = torch.from_numpy(z['x']).requires_grad_(True)
        t0 = time.time()
        out = M.run(ops, model, x)
        (out * torch.from_numpy(z['cot'])).sum().backward()
        e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
        e_dx = rel_l2(x.grad.numpy(), z['dx'])
        bad = {n: round(per[n], 4) for n in names if per[n] > max(3 * ac_per[n], 0.08)}
        ok = e_all < min(2 * ac_glob, max(0.08, ac_glob)) and not bad
        print(f'seed {pseed} {tag:24s}: out {rel_l2(out.detach().numpy(), z["out"]):.4f} dx {e_dx:.4f} grad_global {e_all:.4f} (reference under autocast '
              f'{ac_glob:.4f}) worst {worst} {e_worst:.4f}  gates {"PASS" if ok else "FAIL " + str(bad)}'
              f'{"  [" + str(len(nraw)) + " LayerNorms raw]" if nraw is not None else ""}  [{time.time() - t0:.0f}s]', flush=True)
