"""Per-kernel parity on a real MI355X: every C-ABI entry of libmbx.so against a plain PyTorch fp32
reference of the same op (tests/mock_ops.py, run on the GPU), same seeded inputs.

Tolerances (relative L2 over the whole tensor):
  fp32 mode  : 2e-5   (exact fp32 MFMA; only the summation order differs)
  bf16 mode  : fp32 outputs 2e-5 (operands are the same bf16 values, products exact in fp32);
               bf16 outputs 4e-3 (one rounding of the result); attention 1.5e-2 (P / dS are rounded
               to bf16 before the second MFMA, as in any bf16 flash attention)
Every measured error is also appended to gpurun_out/kernel_parity.json for the round report."""
import json
import os

import pytest
import torch

from motionbert_amd.engine import (EPI_DGELU, EPI_GELU, EPI_RESID, EPI_STORE, EPI_TANH, MODE_SPATIAL, MODE_TEMPORAL)
from tests.mock_ops import MockOps

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REPORT = {}


@pytest.fixture(scope='module')
def ops():
    from motionbert_amd import hip_ops
    return hip_ops.get()


@pytest.fixture(scope='module', autouse=True)
def _dump_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'kernel_parity.json'), 'w') as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def rel(a, b, floor=0.0):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(max(floor, 1e-30)))


def check(name, got, ref, tol, floor=0.0):
    """relative L2 error; `floor` is an absolute lower bound for the denominator (used where the exact
    answer is identically zero, e.g. dq/dk of a one-key softmax)."""
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all(), f'{name}: non-finite output'
    e = rel(got.float(), ref.float(), floor)
    REPORT[name] = e
    assert e < tol, f'{name}: rel-l2 {e:.3e} >= {tol:.1e}'


def rnd(*shape, seed=0, dtype=torch.float32, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


TD = [torch.float32, torch.bfloat16]
TOL_T = {torch.float32: 2e-5, torch.bfloat16: 4e-3}


def tname(dt):
    return 'f32' if dt == torch.float32 else 'bf16'


# ---------------------------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize('C', [64, 256, 512])
def test_embed_fwd_bwd(ops, C):
    B, T, J = 2, 9, 17
    M = B * T * J
    x, w, b = rnd(B, T, J, 3, seed=1), rnd(C, 3, seed=2), rnd(C, seed=3)
    pos, temp = rnd(1, J, C, seed=4), rnd(1, 16, 1, C, seed=5)
    h, h_ref = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    ops.embed_fwd(x, w, b, pos, temp, h, B, T, J)
    MockOps().embed_fwd(x, w, b, pos, temp, h_ref, B, T, J)
    check(f'embed_fwd.C{C}', h, h_ref, 1e-6)
    dh = rnd(M, C, seed=6)
    outs = [[torch.full(s, 7.0, device=DEV) for s in [(C, 3), (C,), (1, J, C), (1, 16, 1, C), (B, T, J, 3)]] for _ in range(2)]
    ops.embed_bwd(dh, x, w, *outs[0], B, T, J)
    MockOps().embed_bwd(dh, x, w, *outs[1], B, T, J)
    for n, a, r in zip(['dw', 'db', 'dpos', 'dtemp', 'dx'], *outs):
        check(f'embed_bwd.{n}.C{C}', a, r, 2e-5)
    # the incoming gradient as a bf16 pair (round 5: the two Blocks' input gradients of level 0): bit for bit embed_bwd on their fp32 sum
    dh_a, dh_b = rnd(M, C, seed=7, dtype=torch.bfloat16), rnd(M, C, seed=8, dtype=torch.bfloat16, scale=0.3)
    pair = [torch.full_like(t, 7.0) for t in outs[0]]
    ops.embed_bwd_pair(dh_a, dh_b, x, w, *pair, B, T, J)
    ops.embed_bwd(dh_a.float() + dh_b.float(), x, w, *outs[0], B, T, J)
    MockOps().embed_bwd_pair(dh_a, dh_b, x, w, *outs[1], B, T, J)
    torch.cuda.synchronize()
    for n, a, o, r in zip(['dw', 'db', 'dpos', 'dtemp', 'dx'], pair, *outs):
        assert torch.equal(a, o), n
        check(f'embed_bwd_pair.{n}.C{C}', a, r, 2e-5)


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('M,C', [(306, 64), (4131, 256), (4131, 512), (1000, 1024)])
def test_layernorm(ops, dt, M, C):
    x, g, b = rnd(M, C, seed=1, scale=2.0) + 0.5, rnd(C, seed=2) * 0.3 + 1, rnd(C, seed=3) * 0.1
    y, mean, rstd = torch.empty(M, C, device=DEV, dtype=dt), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    y2, mean2, rstd2 = torch.empty_like(y), torch.empty_like(mean), torch.empty_like(rstd)
    ops.layernorm_fwd(x, g, b, 1e-6, y, mean, rstd)
    MockOps().layernorm_fwd(x, g, b, 1e-6, y2, mean2, rstd2)
    tag = f'{tname(dt)}.M{M}.C{C}'
    check(f'ln_fwd.y.{tag}', y, y2, TOL_T[dt])
    check(f'ln_fwd.mean.{tag}', mean, mean2, 1e-5)
    check(f'ln_fwd.rstd.{tag}', rstd, rstd2, 1e-5)
    dy = rnd(M, C, seed=4, dtype=dt)
    dres, extra = rnd(M, C, seed=5), rnd(M, C, seed=6)
    for variant, (dr, ex, want_t) in {'full': (dres, extra, True), 'bare': (None, None, False)}.items():
        o = [[torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV, dtype=dt) if want_t else None,
              torch.empty(C, device=DEV), torch.empty(C, device=DEV)] for _ in range(2)]
        ops.layernorm_bwd(dy, x, mean2, rstd2, g, dr, ex, *o[0])
        MockOps().layernorm_bwd(dy, x, mean2, rstd2, g, dr, ex, *o[1])
        check(f'ln_bwd.dx.{variant}.{tag}', o[0][0], o[1][0], 2e-5)
        if want_t:
            check(f'ln_bwd.dx_t.{variant}.{tag}', o[0][1], o[1][1], TOL_T[dt])
        check(f'ln_bwd.dg.{variant}.{tag}', o[0][2], o[1][2], 5e-5)
        check(f'ln_bwd.db.{variant}.{tag}', o[0][3], o[1][3], 5e-5)


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('M,C', [(306, 64), (4131, 512)])
def test_fuse(ops, dt, M, C):
    x_st, x_ts = rnd(M, C, seed=1), rnd(M, C, seed=2)
    w, b = rnd(2, 2 * C, seed=3, scale=0.05), rnd(2, seed=4)
    o = [[torch.empty(M, C, device=DEV), torch.empty(M, 2, device=DEV)] for _ in range(2)]
    ops.fuse_fwd(x_st, x_ts, w, b, *o[0])
    MockOps().fuse_fwd(x_st, x_ts, w, b, *o[1])
    tag = f'{tname(dt)}.M{M}.C{C}'
    check(f'fuse_fwd.out.{tag}', o[0][0], o[1][0], 1e-5)
    check(f'fuse_fwd.alpha.{tag}', o[0][1], o[1][1], 1e-5)
    # the fusion kernel that also emits the LayerNorm(s) of its output (next level's norm1_s / norm1_t, or the final norm)
    g1, b1, g2, b2 = rnd(C, seed=7) * 0.3 + 1, rnd(C, seed=8) * 0.1, rnd(C, seed=9) * 0.3 + 1, rnd(C, seed=10) * 0.1
    for two in (True, False):
        f = [[torch.empty(M, C, device=DEV), torch.empty(M, 2, device=DEV), torch.empty(M, C, device=DEV, dtype=dt),
              torch.empty(M, C, device=DEV, dtype=dt) if two else None, torch.empty(M, device=DEV), torch.empty(M, device=DEV)] for _ in range(2)]
        for impl, r in ((ops, f[0]), (MockOps(), f[1])):
            impl.fuse_ln_fwd(x_st, x_ts, w, b, r[0], r[1], g1, b1, r[2], g2 if two else None, b2 if two else None, r[3], 1e-6, r[4], r[5])
        for n, u, v, tol in zip(['out', 'alpha', 'xn1', 'xn2', 'mean', 'rstd'], f[0], f[1], [1e-5, 1e-5, TOL_T[dt], TOL_T[dt], 2e-5, 2e-5]):
            if u is not None:
                check(f'fuse_ln_fwd.{n}.{"two" if two else "one"}.{tag}', u, v, tol)
        assert torch.equal(f[0][0], o[0][0]) and torch.equal(f[0][1], o[0][1])      # same fused row as the plain kernel, bit for bit
    dh, alpha = rnd(M, C, seed=5), o[1][1]
    mk = lambda: [torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV, dtype=dt),
                  torch.empty(M, C, device=DEV, dtype=dt), torch.empty(2, 2 * C, device=DEV), torch.empty(2, device=DEV)]
    a, r = mk(), mk()
    ops.fuse_bwd(dh, x_st, x_ts, alpha, w, *a)
    MockOps().fuse_bwd(dh, x_st, x_ts, alpha, w, *r)
    for n, u, v, tol in zip(['d_st', 'd_ts', 'd_st_t', 'd_ts_t', 'dw', 'db'], a, r, [2e-5, 2e-5, TOL_T[dt], TOL_T[dt], 5e-5, 5e-5]):
        check(f'fuse_bwd.{n}.{tag}', u, v, tol)
    n = M * C
    avg, avg_r = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    ops.average(x_st, x_ts, avg)
    MockOps().average(x_st, x_ts, avg_r)
    check(f'average.{tag}', avg, avg_r, 1e-6)
    a, r = mk()[:4], mk()[:4]
    ops.average_bwd(dh, *a)
    MockOps().average_bwd(dh, *r)
    for nme, u, v in zip(['d_st', 'd_ts', 'd_st_t', 'd_ts_t'], a, r):
        check(f'average_bwd.{nme}.{tag}', u, v, TOL_T[dt] if 't' in nme[-2:] else 1e-6)


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('M,R,D', [(306, 64, 3), (4131, 512, 3), (500, 512, 8)])
def test_head(ops, dt, M, R, D):
    rep, w, b = torch.tanh(rnd(M, R, seed=1)), rnd(D, R, seed=2, scale=0.1), rnd(D, seed=3)
    out, out_r = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV)
    ops.head_fwd(rep, w, b, out)
    MockOps().head_fwd(rep, w, b, out_r)
    tag = f'{tname(dt)}.M{M}.R{R}.D{D}'
    check(f'head_fwd.{tag}', out, out_r, 1e-5)
    dout = rnd(M, D, seed=4)
    mk = lambda: [torch.empty(M, R, device=DEV, dtype=dt), torch.empty(D, R, device=DEV), torch.empty(D, device=DEV)]
    a, r = mk(), mk()
    ops.head_bwd(dout, rep, w, *a)
    MockOps().head_bwd(dout, rep, w, *r)
    for n, u, v, tol in zip(['dpre', 'dw', 'db'], a, r, [TOL_T[dt], 5e-5, 5e-5]):
        check(f'head_bwd.{n}.{tag}', u, v, tol)
    drep = rnd(M, R, seed=5)
    t1, t2 = torch.empty(M, R, device=DEV, dtype=dt), torch.empty(M, R, device=DEV, dtype=dt)
    ops.tanh_bwd(drep, rep, t1)
    MockOps().tanh_bwd(drep, rep, t2)
    check(f'tanh_bwd.{tag}', t1, t2, TOL_T[dt])


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('numel', [64, 4131 * 1024, 1000 * 384 + 4])
def test_gelu_fwd(ops, dt, numel):
    """mbx_gelu_fwd (recompute mode rebuilds the MLP post-activation with it, engine._mlp_bwd) against the exact erf form
    nn.GELU() computes (DSTformer.py:70), incl. a size that is no multiple of the block size and the tails |u| > 6; an element
    count that is no multiple of 4 (never the case for [M, hidden]) is refused loudly."""
    u = rnd(numel, seed=11, dtype=dt, scale=2.0)
    u[:8] = torch.tensor([-9.0, -6.0, -0.75, -0.0, 0.0, 0.75, 6.0, 9.0], device=DEV, dtype=dt)
    g = torch.full((numel + 16,), 7.0, device=DEV, dtype=dt)      # guard band behind the output
    ops.gelu_fwd(u, g[:numel])
    ref = torch.nn.functional.gelu(u.float())
    check(f'gelu_fwd.{tname(dt)}.{numel}', g[:numel], ref, TOL_T[dt])
    assert float((g[:numel].float() - ref).abs().max()) < (2e-6 if dt == torch.float32 else 0.04)
    assert bool((g[numel:] == 7.0).all()), 'gelu_fwd wrote past its output'
    with pytest.raises(RuntimeError, match='gelu_fwd'):
        ops.gelu_fwd(u[:numel - 1], g[:numel - 1])


@pytest.mark.parametrize('dt', TD)
def test_prep_weights(ops, dt):
    P = {'a.weight': rnd(192, 64, seed=1), 'b.weight': rnd(64, 128, seed=2), 'c.weight': rnd(1536, 512, seed=3)}
    Wn, Wt = ops.prep_weights(P, ['a', 'b', 'c'], dt, True)
    torch.cuda.synchronize()
    for n in 'abc':
        w = P[n + '.weight']
        assert torch.equal(Wn[n], w.to(dt)), n
        assert torch.equal(Wt[n], w.t().contiguous().to(dt)), n
    REPORT[f'prep_weights.{tname(dt)}'] = 0.0


# ---------------------------------------------------------------------------------------------- GEMMs
NT_SHAPES = [(306, 64, 64), (306, 192, 64), (306, 64, 192), (306, 128, 64), (4131, 1536, 512), (4131, 512, 512),
             (4131, 1024, 512), (4131, 512, 1024), (4131, 512, 1536), (1000, 256, 256), (129, 768, 256),
             (70227, 512, 512)]     # the last one: several rounds of 256 x 256 tiles with a ragged last row tile


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('M,N,K', NT_SHAPES)
def test_gemm_nt_store(ops, dt, M, N, K):
    a, w, bias = rnd(M, K, seed=1, dtype=dt), rnd(N, K, seed=2, dtype=dt, scale=0.05), rnd(N, seed=3)
    out, ref = torch.empty(M, N, device=DEV, dtype=dt), torch.empty(M, N, device=DEV, dtype=dt)
    ops.gemm_nt(a, w, bias, EPI_STORE, out_t=out)
    MockOps().gemm_nt(a, w, bias, EPI_STORE, out_t=ref)
    check(f'gemm_nt.store.{tname(dt)}.{M}x{N}x{K}', out, ref, TOL_T[dt])
    ops.gemm_nt(a, w, None, EPI_STORE, out_t=out)
    MockOps().gemm_nt(a, w, None, EPI_STORE, out_t=ref)
    check(f'gemm_nt.store_nobias.{tname(dt)}.{M}x{N}x{K}', out, ref, TOL_T[dt])


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('M,N,K', [(306, 128, 64), (4131, 1024, 512), (4131, 512, 1024), (70227, 512, 512)])
def test_gemm_nt_epilogues(ops, dt, M, N, K):
    a, w, bias = rnd(M, K, seed=1, dtype=dt), rnd(N, K, seed=2, dtype=dt, scale=0.05), rnd(N, seed=3)
    tag = f'{tname(dt)}.{M}x{N}x{K}'
    mk_t = lambda: torch.empty(M, N, device=DEV, dtype=dt)
    mk_f = lambda: torch.empty(M, N, device=DEV)
    u, g, u2, g2 = mk_t(), mk_t(), mk_t(), mk_t()
    ops.gemm_nt(a, w, bias, EPI_GELU, out_t=u, out2_t=g)
    MockOps().gemm_nt(a, w, bias, EPI_GELU, out_t=u2, out2_t=g2)
    check(f'gemm_nt.gelu.u.{tag}', u, u2, TOL_T[dt])
    check(f'gemm_nt.gelu.g.{tag}', g, g2, TOL_T[dt])
    resid = rnd(M, N, seed=4)
    y, y2 = mk_f(), mk_f()
    ops.gemm_nt(a, w, bias, EPI_RESID, out_f=y, resid=resid)
    MockOps().gemm_nt(a, w, bias, EPI_RESID, out_f=y2, resid=resid)
    check(f'gemm_nt.resid.{tag}', y, y2, 2e-5)
    ops.gemm_nt(a, w, bias, EPI_TANH, out_f=y)
    MockOps().gemm_nt(a, w, bias, EPI_TANH, out_f=y2)
    check(f'gemm_nt.tanh.{tag}', y, y2, 2e-5)
    aux = rnd(M, N, seed=5, dtype=dt)
    d, d2 = mk_t(), mk_t()
    ops.gemm_nt(a, w, None, EPI_DGELU, out_t=d, aux_t=aux)
    MockOps().gemm_nt(a, w, None, EPI_DGELU, out_t=d2, aux_t=aux)
    check(f'gemm_nt.dgelu.{tag}', d, d2, TOL_T[dt])


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('M,N,K', [(306, 64, 64), (306, 192, 64), (306, 64, 128), (4131, 1536, 512), (4131, 512, 1024),
                                   (4131, 1024, 512), (70227, 512, 512), (33, 64, 64), (1000, 128, 256), (64, 256, 128), (79, 128, 128)])
def test_gemm_tn(ops, dt, M, N, K):
    dy, a = rnd(M, N, seed=1, dtype=dt), rnd(M, K, seed=2, dtype=dt)
    dw, db, dw2, db2 = (torch.empty(N, K, device=DEV), torch.empty(N, device=DEV), torch.empty(N, K, device=DEV),
                        torch.empty(N, device=DEV))
    ops.gemm_tn(dy, a, dw, db)
    MockOps().gemm_tn(dy, a, dw2, db2)
    tag = f'{tname(dt)}.{M}x{N}x{K}'
    check(f'gemm_tn.dw.{tag}', dw, dw2, 3e-5)
    check(f'gemm_tn.db.{tag}', db, db2, 3e-5)
    ops.gemm_tn(dy, a, dw, None)
    check(f'gemm_tn.dw_nobias.{tag}', dw, dw2, 3e-5)


# ---------------------------------------------------------------------------------------------- bf16x3 (fp32-class) GEMMs
@pytest.mark.parametrize('M,N,K', [(306, 64, 64), (306, 192, 128), (4131, 1536, 512), (4131, 512, 1024), (4131, 512, 1536),
                                   (129, 768, 256), (1000, 256, 256)])
def test_gemm_nt_x3(ops, M, N, K):
    """precision 'bf16x3': split-operand GEMM (three bf16 MFMA passes over hi / lo planes) against the EXACT fp32 product
    in fp64 -- the point of the mode is fp32-class accuracy (gate 2e-5; plain bf16 operands sit at 3e-3)."""
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    ah, al = ops.split(a)
    assert float(((ah.float() + al.float()) - a).abs().max() / a.abs().max()) < 2 ** -15
    wp = ops.split(w)
    exact = (a.double() @ w.double().t() + bias.double()).float()
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt((ah, al), wp, bias, EPI_STORE, out_t=out)
    check(f'gemm_nt_x3.store.{M}x{N}x{K}', out, exact, 2e-5)
    ref = torch.empty(M, N, device=DEV)
    MockOps().gemm_nt((ah, al), wp, bias, EPI_STORE, out_t=ref)
    check(f'gemm_nt_x3.store_vs_restatement.{M}x{N}x{K}', out, ref, 2e-6)
    u, g, u2, g2 = (torch.empty(M, N, device=DEV) for _ in range(4))
    ops.gemm_nt((ah, al), wp, bias, EPI_GELU, out_t=u, out2_t=g)
    MockOps().gemm_nt((ah, al), wp, bias, EPI_GELU, out_t=u2, out2_t=g2)
    check(f'gemm_nt_x3.gelu.u.{M}x{N}x{K}', u, u2, 2e-6)
    check(f'gemm_nt_x3.gelu.g.{M}x{N}x{K}', g, g2, 2e-5)
    resid, y, y2 = rnd(M, N, seed=4), torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm_nt((ah, al), wp, bias, EPI_RESID, out_f=y, resid=resid)
    MockOps().gemm_nt((ah, al), wp, bias, EPI_RESID, out_f=y2, resid=resid)
    check(f'gemm_nt_x3.resid.{M}x{N}x{K}', y, y2, 2e-6)
    ops.gemm_nt((ah, al), wp, bias, EPI_TANH, out_f=y)
    MockOps().gemm_nt((ah, al), wp, bias, EPI_TANH, out_f=y2)
    check(f'gemm_nt_x3.tanh.{M}x{N}x{K}', y, y2, 2e-5)
    aux, d, d2 = rnd(M, N, seed=5), torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.gemm_nt((ah, al), wp, None, EPI_DGELU, out_t=d, aux_t=aux)
    MockOps().gemm_nt((ah, al), wp, None, EPI_DGELU, out_t=d2, aux_t=aux)
    check(f'gemm_nt_x3.dgelu.{M}x{N}x{K}', d, d2, 2e-5)
    # the same outputs as operand planes straight from the epilogue: bit for bit the split of the fp32 output
    BF = torch.bfloat16
    pl = lambda: (torch.full((M, N), float('nan'), device=DEV, dtype=BF), torch.full((M, N), float('nan'), device=DEV, dtype=BF))
    same = lambda p, t: all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(p, ops.split(t)))
    p = pl()
    ops.gemm_nt((ah, al), wp, bias, EPI_STORE, out_t=p)
    assert same(p, out), 'store planes'
    p, u3 = pl(), torch.empty(M, N, device=DEV)
    ops.gemm_nt((ah, al), wp, bias, EPI_GELU, out_t=u3, out2_t=p)
    assert same(p, g) and torch.equal(u3, u), 'gelu planes'
    p = pl()
    ops.gemm_nt((ah, al), wp, bias, EPI_GELU, out_t=None, out2_t=p)
    assert same(p, g), 'gelu planes without the pre-activation'
    p = pl()
    ops.gemm_nt((ah, al), wp, None, EPI_DGELU, out_t=p, aux_t=aux)
    assert same(p, d), 'dgelu planes'


@pytest.mark.parametrize('M,C', [(306, 64), (4131, 512), (1000, 256), (77, 1024)])
@pytest.mark.parametrize('affine', [False, True])
def test_layernorm_fwd_planes(ops, M, C, affine):
    """mbx_layernorm_fwd_planes: the bf16x3 operand planes of LayerNorm(x) = the split of the fp32 kernel's output, bit for bit."""
    x = rnd(M, C, seed=1) * 2.0 + 0.3
    g, b = (rnd(C, seed=2) * 0.2 + 1.0, rnd(C, seed=3) * 0.1) if affine else (None, None)
    y, mean, rstd = torch.empty(M, C, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, g, b, 1e-6, y, mean, rstd)
    BF = torch.bfloat16
    p = (torch.full((M, C), float('nan'), device=DEV, dtype=BF), torch.full((M, C), float('nan'), device=DEV, dtype=BF))
    mean2, rstd2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, g, b, 1e-6, p, mean2, rstd2)
    hi, lo = ops.split(y)
    assert torch.equal(p[0].view(torch.int16), hi.view(torch.int16)) and torch.equal(p[1].view(torch.int16), lo.view(torch.int16))
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    if affine:      # backward: the T copy of dx as operand planes = the split of dx, bit for bit; everything else unchanged
        dy, dres = rnd(M, C, seed=4), rnd(M, C, seed=5)
        dx, dg, db = torch.empty(M, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        dx2, dg2, db2 = torch.empty(M, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ops.layernorm_bwd(dy, x, mean, rstd, g, dres, None, dx, None, dg, db)
        q = (torch.full((M, C), float('nan'), device=DEV, dtype=BF), torch.full((M, C), float('nan'), device=DEV, dtype=BF))
        ops.layernorm_bwd(dy, x, mean, rstd, g, dres, None, dx2, q, dg2, db2)
        hi, lo = ops.split(dx)
        assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
        assert torch.equal(q[0].view(torch.int16), hi.view(torch.int16)) and torch.equal(q[1].view(torch.int16), lo.view(torch.int16))


@pytest.mark.parametrize('M,N,K', [(306, 64, 64), (306, 192, 128), (4131, 1536, 512), (4131, 512, 1024), (4131, 512, 512), (70227, 512, 512), (33, 64, 64)])
def test_gemm_tn_x3(ops, M, N, K):
    dy, a = rnd(M, N, seed=1), rnd(M, K, seed=2)
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ops.gemm_tn(ops.split(dy), ops.split(a), dw, db)
    check(f'gemm_tn_x3.dw.{M}x{N}x{K}', dw, (dy.double().t() @ a.double()).float(), 3e-5)
    check(f'gemm_tn_x3.db.{M}x{N}x{K}', db, dy.double().sum(0).float(), 3e-5)
    ops.gemm_tn(ops.split(dy), ops.split(a), dw, None)
    check(f'gemm_tn_x3.dw_nobias.{M}x{N}x{K}', dw, (dy.double().t() @ a.double()).float(), 3e-5)


# ---------------------------------------------------------------------------------------------- attention
ATT = [(2, 9, 2, 32), (2, 9, 8, 64), (3, 30, 8, 32), (2, 81, 8, 32), (2, 81, 8, 64), (1, 243, 8, 64), (1, 243, 8, 32),
       (2, 100, 2, 64), (1, 1, 8, 64), (5, 33, 8, 64)]


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('mode', [MODE_SPATIAL, MODE_TEMPORAL])
@pytest.mark.parametrize('B,T,H,hd', ATT)
def test_attention(ops, dt, mode, B, T, H, hd):
    J, C = 17, H * hd
    M = B * T * J
    qkv = rnd(M, 3 * C, seed=1, dtype=dt)
    # make the softmax non-trivial: a few dominant keys (forces large row maxima)
    qkv[:, :C] *= 2.0
    scale = hd ** -0.5
    tol_o = 2e-5 if dt == torch.float32 else 1.5e-2
    tag = f'{tname(dt)}.{"sp" if mode == MODE_SPATIAL else "tm"}.B{B}T{T}H{H}d{hd}'
    o, lse = torch.full((M, C), 9.0, device=DEV, dtype=dt), torch.full((M, H), 9.0, device=DEV)
    o2, lse2 = torch.empty(M, C, device=DEV, dtype=dt), torch.empty(M, H, device=DEV)
    ops.attn_fwd(qkv, o, lse, B, T, J, H, scale, mode)
    MockOps().attn_fwd(qkv, o2, lse2, B, T, J, H, scale, mode)
    check(f'attn_fwd.o.{tag}', o, o2, tol_o)
    check(f'attn_fwd.lse.{tag}', lse, lse2, 2e-5 if dt == torch.float32 else 1e-4)
    do = rnd(M, C, seed=2, dtype=dt)
    dq, dq2 = torch.full((M, 3 * C), 9.0, device=DEV, dtype=dt), torch.empty(M, 3 * C, device=DEV, dtype=dt)
    ops.attn_bwd(qkv, o2, do, lse2, dq, B, T, J, H, scale, mode)
    MockOps().attn_bwd(qkv, o2, do, lse2, dq2, B, T, J, H, scale, mode)
    # L == 1: p == 1, dS == 0 exactly -> dq = dk = 0; judge those against the scale of dv instead of 0
    floor = float(dq2[:, 2 * C:].float().norm()) if (T if mode == MODE_TEMPORAL else J) == 1 else 0.0
    for i, n in enumerate(['dq', 'dk', 'dv']):
        check(f'attn_bwd.{n}.{tag}', dq[:, i * C:(i + 1) * C], dq2[:, i * C:(i + 1) * C], 5e-5 if dt == torch.float32 else 2e-2, floor)
    if dt == torch.float32:      # the same gradients as the operand planes of the bf16x3 split: bit for bit the split of the fp32 output
        BF = torch.bfloat16
        pl = (torch.full((M, 3 * C), 9.0, device=DEV, dtype=BF), torch.full((M, 3 * C), 9.0, device=DEV, dtype=BF))
        ops.attn_bwd(qkv, o2, do, lse2, pl, B, T, J, H, scale, mode)
        hi, lo = ops.split(dq)
        assert torch.equal(pl[0].view(torch.int16), hi.view(torch.int16)) and torch.equal(pl[1].view(torch.int16), lo.view(torch.int16)), tag


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('mode,B,T,H,hd', [(MODE_SPATIAL, 3, 9, 4, 32), (MODE_SPATIAL, 2, 5, 8, 64), (MODE_TEMPORAL, 2, 27, 4, 64),
                                           (MODE_TEMPORAL, 2, 81, 4, 32), (MODE_TEMPORAL, 1, 243, 8, 64), (MODE_TEMPORAL, 2, 100, 2, 64)])
def test_attention_probability_dropout(ops, dt, mode, B, T, H, hd):
    """nn.Dropout on the softmax output (attn_drop, DSTformer.py:96,182,196) INSIDE the attention kernels: forward and backward
    with the counter-based mask against the torch restatement that multiplies the materialised probabilities by
    dropmask.mask_like over the reference's attn tensor layout (and that tests/test_host_logic.py pins to the real reference
    run with forced masks).  Every kernel family: one wave per short problem, shared tiles for long ones, the dQ + dK/dV pair."""
    J, C = 17, H * hd
    M = B * T * J
    qkv = rnd(M, 3 * C, seed=1, dtype=dt)
    qkv[:, :C] *= 2.0
    scale = hd ** -0.5
    drop = (0.25, 0x1234567890ABCDEF)
    tol_o = 2e-5 if dt == torch.float32 else 1.5e-2
    tag = f'{tname(dt)}.{"sp" if mode == MODE_SPATIAL else "tm"}.B{B}T{T}H{H}d{hd}'
    o, lse = torch.full((M, C), 9.0, device=DEV, dtype=dt), torch.full((M, H), 9.0, device=DEV)
    o2, lse2 = torch.empty(M, C, device=DEV, dtype=dt), torch.empty(M, H, device=DEV)
    o0 = torch.empty(M, C, device=DEV, dtype=dt)
    ops.attn_fwd(qkv, o, lse, B, T, J, H, scale, mode, drop=drop)
    ops.attn_fwd(qkv, o0, torch.empty_like(lse), B, T, J, H, scale, mode)
    MockOps().attn_fwd(qkv, o2, lse2, B, T, J, H, scale, mode, drop=drop)
    check(f'attn_drop.fwd.o.{tag}', o, o2, tol_o)
    check(f'attn_drop.fwd.lse.{tag}', lse, lse2, 2e-5 if dt == torch.float32 else 1e-4)
    assert rel(o.float(), o0.float()) > 0.1, 'the mask must change the output'
    do = rnd(M, C, seed=2, dtype=dt)
    dq, dq2 = torch.full((M, 3 * C), 9.0, device=DEV, dtype=dt), torch.empty(M, 3 * C, device=DEV, dtype=dt)
    ops.attn_bwd(qkv, o2, do, lse2, dq, B, T, J, H, scale, mode, drop=drop)
    MockOps().attn_bwd(qkv, o2, do, lse2, dq2, B, T, J, H, scale, mode, drop=drop)
    for i, n in enumerate(['dq', 'dk', 'dv']):
        check(f'attn_drop.bwd.{n}.{tag}', dq[:, i * C:(i + 1) * C], dq2[:, i * C:(i + 1) * C], 5e-5 if dt == torch.float32 else 2e-2)


# ---------------------------------------------------------------------------------------------- memory safety
class Guarded:
    """Output/workspace buffers carved out of a larger allocation with sentinel bands on both sides: a kernel
    that stores outside the tensor it was handed (ragged last tile, rounded-up grid) trips `verify()`."""
    BAND = 1 << 18   # elements on each side

    def __init__(self):
        self.items = []

    def __call__(self, *shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= s
        raw = torch.empty(n + 2 * self.BAND, dtype=torch.uint8 if dtype == torch.uint8 else dtype, device=DEV)
        sentinel = 85 if dtype == torch.uint8 else 12345.0
        raw.fill_(sentinel)
        self.items.append((raw, n, sentinel))
        return raw[self.BAND:self.BAND + n].view(*shape)

    def verify(self, what):
        torch.cuda.synchronize()
        for raw, n, sentinel in self.items:
            lo, hi = raw[:self.BAND], raw[self.BAND + n:]
            assert bool((lo == sentinel).all()) and bool((hi == sentinel).all()), \
                f'{what}: store outside a {n}-element {raw.dtype} buffer'
        self.items.clear()


@pytest.mark.parametrize('dt', TD)
@pytest.mark.parametrize('B,T', [(16, 243), (3, 50), (1, 1)])
def test_no_kernel_stores_outside_its_buffers(ops, dt, B, T, monkeypatch):
    """Every entry of the ABI at training shapes (ragged M: 16*243*17 = 66096 = 258*256 + 48) with guarded outputs
    and guarded workspaces."""
    J, C, H, HID, R = 17, 512, 8, 1024, 512
    M = B * T * J
    G = Guarded()
    monkeypatch.setattr(ops, '_ws', lambda key, fn, *a, device=None: G(max(int(fn(*a)), 16), dtype=torch.uint8))
    torch.manual_seed(B * 1000 + T)
    f = lambda *s: torch.randn(*s, device=DEV) * 0.5
    t = lambda *s: f(*s).to(dt)
    tag = f'{tname(dt)}.B{B}T{T}'
    # embedding
    x, w, b, pos, temp = f(B, T, J, 3), f(C, 3), f(C), f(1, J, C), f(1, 243, 1, C)
    ops.embed_fwd(x, w, b, pos, temp, G(M, C), B, T, J)
    G.verify(f'embed_fwd {tag}')
    ops.embed_bwd(f(M, C), x, w, G(C, 3), G(C), G(1, J, C), G(1, 243, 1, C), G(B, T, J, 3), B, T, J)
    G.verify(f'embed_bwd {tag}')
    # LayerNorm
    xs, gam, bet = f(M, C), f(C), f(C)
    mean, rstd = G(M), G(M)
    ops.layernorm_fwd(xs, gam, bet, 1e-6, G(M, C, dtype=dt), mean, rstd)
    G.verify(f'layernorm_fwd {tag}')
    mean, rstd = xs.mean(1), 1.0 / (xs.var(1, unbiased=False) + 1e-6).sqrt()
    for dx_t in (True, False):
        ops.layernorm_bwd(t(M, C), xs, mean, rstd, gam, f(M, C), None, G(M, C), G(M, C, dtype=dt) if dx_t else None, G(C), G(C))
        G.verify(f'layernorm_bwd {tag} dx_t={dx_t}')
    # GEMMs at the shapes of the model
    for N, K in [(3 * C, C), (C, C), (HID, C), (C, HID), (R, C)]:
        a, wt, bias = t(M, K), t(N, K), f(N)
        ops.gemm_nt(a, wt, bias, EPI_STORE, out_t=G(M, N, dtype=dt))
        G.verify(f'gemm_nt store {tag} N{N} K{K}')
        ops.gemm_nt(a, wt, bias, EPI_GELU, out_t=G(M, N, dtype=dt), out2_t=G(M, N, dtype=dt))
        ops.gemm_nt(a, wt, bias, EPI_GELU, out_t=None, out2_t=G(M, N, dtype=dt))
        G.verify(f'gemm_nt gelu {tag} N{N} K{K}')
        ops.gemm_nt(a, wt, bias, EPI_RESID, resid=f(M, N), out_f=G(M, N))
        ops.gemm_nt(a, wt, bias, EPI_TANH, out_f=G(M, N))
        G.verify(f'gemm_nt resid/tanh {tag} N{N} K{K}')
        ops.gemm_nt(a, wt, None, EPI_DGELU, out_t=G(M, N, dtype=dt), aux_t=t(M, N))
        G.verify(f'gemm_nt dgelu {tag} N{N} K{K}')
        ops.gemm_tn(t(M, N), a, G(N, K), G(N))
        ops.gemm_tn(t(M, N), a, G(N, K), None)
        G.verify(f'gemm_tn {tag} N{N} K{K}')
    # attention
    for mode in (MODE_SPATIAL, MODE_TEMPORAL):
        qkv = t(M, 3 * C)
        o, lse = G(M, C, dtype=dt), G(M, H)
        ops.attn_fwd(qkv, o, lse, B, T, J, H, (C // H) ** -0.5, mode)
        G.verify(f'attn_fwd {tag} mode{mode}')
        ops.attn_bwd(qkv, o.clone(), t(M, C), lse.clone(), G(M, 3 * C, dtype=dt), B, T, J, H, (C // H) ** -0.5, mode)
        G.verify(f'attn_bwd {tag} mode{mode}')
    # fusion, tail
    x_st, x_ts, fw, fb = f(M, C), f(M, C), f(2, 2 * C), f(2)
    alpha = G(M, 2)
    ops.fuse_fwd(x_st, x_ts, fw, fb, G(M, C), alpha)
    G.verify(f'fuse_fwd {tag}')
    ops.fuse_ln_fwd(x_st, x_ts, fw, fb, G(M, C), G(M, 2), gam, bet, G(M, C, dtype=dt), gam, bet, G(M, C, dtype=dt), 1e-6, G(M), G(M))
    ops.fuse_ln_fwd(x_st, x_ts, fw, fb, G(M, C), G(M, 2), gam, bet, G(M, C, dtype=dt), None, None, None, 1e-6, G(M), G(M))
    G.verify(f'fuse_ln_fwd {tag}')
    ops.fuse_bwd(f(M, C), x_st, x_ts, alpha.clone(), fw, G(M, C), G(M, C), G(M, C, dtype=dt), G(M, C, dtype=dt), G(2, 2 * C), G(2))
    G.verify(f'fuse_bwd {tag}')
    ops.average(x_st, x_ts, G(M, C))
    ops.average_bwd(f(M, C), G(M, C), G(M, C), G(M, C, dtype=dt), G(M, C, dtype=dt))
    G.verify(f'average {tag}')
    rep, hw, hb = torch.tanh(f(M, R)), f(3, R), f(3)
    ops.head_fwd(rep, hw, hb, G(M, 3))
    ops.head_bwd(f(M, 3), rep, hw, G(M, R, dtype=dt), G(3, R), G(3))
    ops.tanh_bwd(f(M, R), rep, G(M, R, dtype=dt))
    G.verify(f'tail {tag}')
