"""LayerNorm folding (round 3) on a real MI355X: every new C-ABI entry against its torch restatement in oracle/torch_ops.py
(MockOps, run on the GPU with the same bf16 operands), then the two backward formulations of the whole model against each other
and against the reference-minted fixture.

What is being checked is an identity, not an approximation: with W' = W diag(gamma), b' = b + W beta and Y = xhat W'^T + b',
    mean_k(dxhat) = (1/C) dY . rowsum(W'),      mean_k(dxhat xhat) = (1/C) dY . (Y - b'),
so LayerNorm's backward needs no row reduction over the dX GEMM's output (include/mbx.h "LayerNorm folded into the Linear it
feeds").  Tolerances: fp32 outputs of bf16 operands 2e-5 (only the summation order differs); bf16 outputs 4e-3 (one rounding)."""
import pytest
import torch

from motionbert_amd.engine import EPI_DGELU, EPI_GELU, MODE_SPATIAL, MODE_TEMPORAL
from tests.helpers import build_model, load_golden
from tests.mock_ops import MockOps
from tests.test_gpu_kernels import DEV, REPORT, check, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    from motionbert_amd import hip_ops
    return hip_ops.get()


@pytest.fixture(scope='module', autouse=True)
def _dump_report():
    import json
    import os
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'fold_parity.json'), 'w') as f:
        json.dump({k: v for k, v in REPORT.items() if 'fold' in k or 'stats' in k or 'lnbwd' in k or 'plain' in k}, f, indent=1, sort_keys=True)


@pytest.mark.parametrize('bias', [True, False])
def test_fold_norm_weights(ops, bias):
    """W' = bf16(W diag(gamma)) and its transpose, b' = b + W beta, rsum = row sums of the ROUNDED W' -- shapes of the full model
    (qkv 1536 x 512, fc1 1024 x 512) and one that is no multiple of the 32 x 32 tile."""
    P = {}
    shapes = [('q', 1536, 512), ('f', 1024, 512), ('odd', 200, 72)]
    for i, (n, N, K) in enumerate(shapes):
        P[n + '.weight'] = rnd(N, K, seed=10 + i, scale=0.05)
        if bias:
            P[n + '.bias'] = rnd(N, seed=20 + i)
        P['n' + n + '.weight'] = 1.0 + 0.3 * rnd(K, seed=30 + i)
        P['n' + n + '.bias'] = 0.2 * rnd(K, seed=40 + i)
    pairs = [(n, 'n' + n) for n, _, _ in shapes]
    got = ops.fold_norm_weights(P, pairs, True)
    ref = MockOps().fold_norm_weights(P, pairs, True, BF)
    torch.cuda.synchronize()
    for n, _, _ in shapes:
        tag = f'fold_norm_weights.{n}.{"bias" if bias else "nobias"}'
        assert torch.equal(got[0][n], ref[0][n]), f'{tag}: folded weight differs bitwise'
        assert torch.equal(got[1][n], ref[1][n]), f'{tag}: transposed folded weight differs bitwise'
        check(tag + '.bias_f', got[2][n], ref[2][n], 2e-6)
        check(tag + '.rsum', got[3][n], ref[3][n], 2e-6, floor=1e-3)


@pytest.mark.parametrize('M,C', [(4131, 512), (1000, 256), (777, 64)])
def test_plain_normalisation(ops, M, C):
    """mbx_layernorm_fwd / mbx_fuse_ln_fwd with gamma = beta = NULL write xhat = (x - mean) rstd."""
    x = rnd(M, C, seed=1) * 3 + 0.7
    y, mean, rstd = torch.empty(M, C, device=DEV, dtype=BF), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, None, None, 1e-6, y, mean, rstd)
    mu = x.mean(-1, keepdim=True)
    rs = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
    check(f'layernorm_fwd.plain.M{M}.C{C}', y, (x - mu) * rs, 4e-3)
    check(f'layernorm_fwd.plain.rstd.M{M}.C{C}', rstd, rs[:, 0], 1e-5)
    x_st, x_ts, fw, fb = rnd(M, C, seed=2), rnd(M, C, seed=3), rnd(2, 2 * C, seed=4, scale=0.05), rnd(2, seed=5)
    mk = lambda: [torch.empty(M, C, device=DEV), torch.empty(M, 2, device=DEV), torch.empty(M, C, device=DEV, dtype=BF), torch.empty(M, device=DEV),
                  torch.empty(M, device=DEV)]
    a, r = mk(), mk()
    ops.fuse_ln_fwd(x_st, x_ts, fw, fb, a[0], a[1], None, None, a[2], None, None, None, 1e-6, a[3], a[4])
    MockOps().fuse_ln_fwd(x_st, x_ts, fw, fb, r[0], r[1], None, None, r[2], None, None, None, 1e-6, r[3], r[4])
    for n, u, v, tol in zip(['out', 'alpha', 'xhat', 'mean', 'rstd'], a, r, [1e-5, 1e-5, 4e-3, 1e-4, 1e-5]):
        check(f'fuse_ln_fwd.plain.{n}.M{M}.C{C}', u, v, tol, floor=1e-3)


@pytest.mark.parametrize('M,N,K', [(4131, 1024, 512), (1000, 256, 128), (264384 // 16, 1024, 512), (300, 128, 64)])
def test_gemm_nt_dgelu_stats(ops, M, N, K):
    """GELU' epilogue + the row dots of its (rounded) output with rsum and (u - bias_f), per 64-column block."""
    a, w = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, dtype=BF, scale=0.05)
    u = rnd(M, N, seed=3, dtype=BF, scale=1.5)
    bias_f, rsum = rnd(N, seed=4, scale=0.3), rnd(N, seed=5)
    mk = lambda: [torch.empty(M, N, device=DEV, dtype=BF), torch.full((N // 64, M, 2), 7.0, device=DEV)]
    g, r = mk(), mk()
    ops.gemm_nt_dgelu_stats(a, w, g[0], u, bias_f, rsum, g[1])
    MockOps().gemm_nt_dgelu_stats(a, w, r[0], u, bias_f, rsum, r[1])
    tag = f'M{M}.N{N}.K{K}'
    check(f'gemm_nt_dgelu_stats.du.{tag}', g[0], r[0], 4e-3)
    # the dots are taken over the kernel's OWN rounded output: compare against the same dots of that output (exact up to fp32 order)
    d = g[0].float()
    rb, bb = rsum.to(BF).float(), bias_f.to(BF).float()      # the kernel multiplies packed bf16 pairs: both vectors enter rounded
    own = torch.stack([(d * rb).reshape(M, N // 64, 64).sum(-1), (d * (u.float() - bb)).reshape(M, N // 64, 64).sum(-1)], -1).transpose(0, 1)
    check(f'gemm_nt_dgelu_stats.part.{tag}', g[1], own, 1e-4)
    check(f'gemm_nt_dgelu_stats.part_vs_ref.{tag}', g[1], r[1], 2e-2)


@pytest.mark.parametrize('M,C', [(4131, 512), (1000, 256), (33, 64), (16524, 512)])
def test_fuse_bwd_pair(ops, M, C):
    """mbx_fuse_bwd_pair: the fusion backward on dh = dh_a + dh_b (two bf16 tensors: the input gradients of the two Blocks above) is bit for
    bit mbx_fuse_bwd on their fp32 sum, and agrees with the torch restatement."""
    x_st, x_ts = rnd(M, C, seed=1), rnd(M, C, seed=2)
    w, fb = rnd(2, 2 * C, seed=3, scale=0.05), rnd(2, seed=4)
    alpha = torch.softmax(torch.cat([x_st, x_ts], -1) @ w.t() + fb, -1)
    dh_a, dh_b = rnd(M, C, seed=5, dtype=BF), rnd(M, C, seed=6, dtype=BF, scale=0.3)
    mk = lambda: [torch.full((M, C), 7.0, device=DEV, dtype=BF), torch.full((M, C), 7.0, device=DEV, dtype=BF), torch.empty(2, 2 * C, device=DEV),
                  torch.empty(2, device=DEV)]
    g, o, r = mk(), mk(), mk()
    ops.fuse_bwd_pair(dh_a, dh_b, x_st, x_ts, alpha, w, *g)
    ops.fuse_bwd(dh_a.float() + dh_b.float(), x_st, x_ts, alpha, w, None, None, *o)
    MockOps().fuse_bwd_pair(dh_a, dh_b, x_st, x_ts, alpha, w, *r)
    torch.cuda.synchronize()
    for name, u, v, ref, tol in zip(('d_st_t', 'd_ts_t', 'dw', 'db'), g, o, r, (4e-3, 4e-3, 2e-5, 2e-5)):
        assert torch.equal(u, v), name
        check(f'fuse_bwd_pair.{name}.M{M}.C{C}', u, ref, tol)


@pytest.mark.parametrize('M,N,K', [(4131, 1024, 512), (1000, 256, 128), (264384 // 16, 1024, 512), (300, 512, 64), (257, 264, 64)])
def test_gemm_nt_gelu_d_and_mul(ops, M, N, K):
    """fc1 + GELU saving the derivative (mbx_gemm_nt_gelu_d) and the one-multiply backward epilogue (mbx_gemm_nt_mul), against the torch
    restatement; the pair against the GELU' epilogue that works from the saved pre-activation (same gradient up to bf16 rounding)."""
    a, w, bias = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, dtype=BF, scale=0.08), rnd(N, seed=3, scale=0.3)
    mk = lambda: [torch.full((M, N), 7.0, device=DEV, dtype=BF), torch.full((M, N), 7.0, device=DEV, dtype=BF)]
    g, r = mk(), mk()
    ops.gemm_nt_gelu_d(a, w, bias, g[0], g[1])
    MockOps().gemm_nt_gelu_d(a, w, bias, r[0], r[1])
    tag = f'M{M}.N{N}.K{K}'
    check(f'gemm_nt_gelu_d.d.{tag}', g[0], r[0], 4e-3)
    check(f'gemm_nt_gelu_d.g.{tag}', g[1], r[1], 4e-3)
    dy, w2 = rnd(M, 64, seed=4, dtype=BF), rnd(N, 64, seed=5, dtype=BF, scale=0.1)
    du, du_r = torch.full((M, N), 7.0, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_nt_mul(dy, w2, g[0], du)
    MockOps().gemm_nt_mul(dy, w2, g[0], du_r)
    check(f'gemm_nt_mul.{tag}', du, du_r, 4e-3)
    # the old pair: pre-activation saved, GELU' in the backward epilogue
    u, g2, du_old = torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
    ops.gemm_nt(a, w, bias, EPI_GELU, out_t=u, out2_t=g2)
    assert torch.equal(g2, g[1]) or float((g2.float() - g[1].float()).abs().max()) < 0.04      # two erf approximations, both far below bf16 resolution
    ops.gemm_nt(dy, w2, None, EPI_DGELU, out_t=du_old, aux_t=u)
    check(f'gemm_nt_mul.vs_dgelu.{tag}', du, du_old, 1.2e-2)


@pytest.mark.parametrize('hd', [32, 64])
@pytest.mark.parametrize('mode,B,T,J', [(MODE_SPATIAL, 3, 9, 17), (MODE_TEMPORAL, 2, 243, 17), (MODE_TEMPORAL, 3, 81, 5), (MODE_TEMPORAL, 2, 27, 17),
                                        (MODE_SPATIAL, 2, 4, 32)])
def test_attn_bwd_stats(ops, mode, B, T, J, hd):
    """Both bf16 backward kernels (one wave per short problem, sixteen waves per long one): dqkv identical to mbx_attn_bwd, and
    per (token, head) the dots of the rounded dqkv with rsum and (qkv - bias_f), q columns and k + v columns separately."""
    H = 4
    C, M = H * hd, B * T * J
    qkv, do = rnd(M, 3 * C, seed=1, dtype=BF), rnd(M, C, seed=2, dtype=BF)
    o, lse = torch.empty(M, C, device=DEV, dtype=BF), torch.empty(M, H, device=DEV)
    scale = hd ** -0.5
    ops.attn_fwd(qkv, o, lse, B, T, J, H, scale, mode)
    bias_f, rsum = rnd(3 * C, seed=3, scale=0.3), rnd(3 * C, seed=4)
    d0, d1 = torch.empty(M, 3 * C, device=DEV, dtype=BF), torch.empty(M, 3 * C, device=DEV, dtype=BF)
    part = torch.full((2 * H, M, 2), 7.0, device=DEV)
    ops.attn_bwd(qkv, o, do, lse, d0, B, T, J, H, scale, mode)
    ops.attn_bwd_stats(qkv, o, do, lse, d1, bias_f, rsum, part, B, T, J, H, scale, mode)
    torch.cuda.synchronize()
    tag = f'mode{mode}.B{B}.T{T}.J{J}.hd{hd}'
    assert torch.equal(d0, d1), f'attn_bwd_stats.{tag}: dqkv differs from mbx_attn_bwd'
    d = d1.float().reshape(M, 3, H, hd)
    rb, bb = rsum.to(BF).float(), bias_f.to(BF).float()      # packed bf16 products: both vectors enter rounded
    y = (qkv.float() - bb).reshape(M, 3, H, hd)
    t1, t2 = (d * rb.reshape(1, 3, H, hd)).sum(3), (d * y).sum(3)         # [M, 3, H]: per tensor (q, k, v) and head
    own = torch.stack([torch.stack([t1[:, 0], t1[:, 1] + t1[:, 2]], -1), torch.stack([t2[:, 0], t2[:, 1] + t2[:, 2]], -1)], -1)   # [M, H, role, 2]
    check(f'attn_bwd_stats.part.{tag}', part, own.reshape(M, 2 * H, 2).transpose(0, 1), 1e-4)


@pytest.mark.parametrize('M,nb,C', [(4131, 8, 512), (1000, 16, 512), (77, 2, 64), (264384 // 8, 16, 512)])
def test_lnbwd_rowc(ops, M, nb, C):
    part, rstd = rnd(nb, M, 2, seed=1), rnd(M, seed=2).abs() + 0.1
    a, r = torch.full((M + 1, 4), 7.0, device=DEV), torch.empty(M, 4, device=DEV)
    ops.lnbwd_rowc(part, rstd, a[:M], C)
    MockOps().lnbwd_rowc(part, rstd, r, C)
    check(f'lnbwd_rowc.M{M}.nb{nb}', a[:M], r, 1e-5)
    assert bool((a[M] == 7.0).all()), 'lnbwd_rowc wrote past its output'


@pytest.mark.parametrize('with_extra,with_t', [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(4131, 512, 1536), (4131, 512, 1024), (1000, 64, 192), (300, 256, 256), (264384 // 16, 512, 1536)])
def test_gemm_nt_lnbwd(ops, M, N, K, with_extra, with_t):
    """LayerNorm backward as the epilogue of the dX GEMM: dx = dres [+ extra] + rowc.x acc - rowc.y - xhat rowc.z (+ bf16 copy)."""
    a, w = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, dtype=BF, scale=0.05)
    xhat, rowc = rnd(M, N, seed=3, dtype=BF), rnd(M, 4, seed=4)
    dres, extra = rnd(M, N, seed=5), (rnd(M, N, seed=6) if with_extra else None)
    mk = lambda: [torch.full((M + 1, N), 7.0, device=DEV), torch.full((M + 1, N), 7.0, device=DEV, dtype=BF) if with_t else None]
    g, r = mk(), mk()
    ops.gemm_nt_lnbwd(a, w, xhat, rowc, dres, extra, g[0][:M], g[1][:M] if with_t else None)
    MockOps().gemm_nt_lnbwd(a, w, xhat, rowc, dres, extra, r[0][:M], r[1][:M] if with_t else None)
    tag = f'M{M}.N{N}.K{K}.{"x" if with_extra else "-"}{"t" if with_t else "-"}'
    check(f'gemm_nt_lnbwd.dx.{tag}', g[0][:M], r[0][:M], 2e-5)
    assert bool((g[0][M] == 7.0).all()), 'gemm_nt_lnbwd wrote past dx'
    if with_t:
        check(f'gemm_nt_lnbwd.dx_t.{tag}', g[1][:M], r[1][:M], 4e-3)
        assert bool((g[1][M] == 7.0).all()), 'gemm_nt_lnbwd wrote past dx_t'


@pytest.mark.parametrize('with_extra,with_f,with_t', [(False, False, True), (True, True, False), (True, True, True), (False, True, False)])
@pytest.mark.parametrize('M,N,K', [(4131, 512, 1536), (4131, 512, 1024), (1000, 64, 192), (264384 // 16, 512, 1536)])
def test_gemm_nt_lnbwd_bf16_gradient_stream(ops, M, N, K, with_extra, with_f, with_t):
    """mbx_gemm_nt_lnbwd_t (round 4): the same epilogue with dres arriving as bf16 and either output optional -- inside a Block only
    the bf16 dx is written (it is the gradient stream and the next GEMMs' operand at once)."""
    a, w = rnd(M, K, seed=1, dtype=BF), rnd(N, K, seed=2, dtype=BF, scale=0.05)
    xhat, rowc = rnd(M, N, seed=3, dtype=BF), rnd(M, 4, seed=4)
    dres, extra = rnd(M, N, seed=5, dtype=BF), (rnd(M, N, seed=6) if with_extra else None)
    mk = lambda: [torch.full((M + 1, N), 7.0, device=DEV) if with_f else None, torch.full((M + 1, N), 7.0, device=DEV, dtype=BF) if with_t else None]
    g, r = mk(), mk()
    ops.gemm_nt_lnbwd(a, w, xhat, rowc, dres, extra, g[0][:M] if with_f else None, g[1][:M] if with_t else None)
    MockOps().gemm_nt_lnbwd(a, w, xhat, rowc, dres, extra, r[0][:M] if with_f else None, r[1][:M] if with_t else None)
    tag = f'M{M}.N{N}.K{K}.{"x" if with_extra else "-"}{"f" if with_f else "-"}{"t" if with_t else "-"}'
    if with_f:
        check(f'gemm_nt_lnbwd_t.dx.{tag}', g[0][:M], r[0][:M], 2e-5)
        assert bool((g[0][M] == 7.0).all()), 'gemm_nt_lnbwd_t wrote past dx'
    if with_t:
        check(f'gemm_nt_lnbwd_t.dx_t.{tag}', g[1][:M], r[1][:M], 4e-3)
        assert bool((g[1][M] == 7.0).all()), 'gemm_nt_lnbwd_t wrote past dx_t'


@pytest.mark.parametrize('N,K', [(1536, 512), (1024, 512), (192, 64), (200, 72)])
def test_unfold_norm_grads(ops, N, K):
    dw, db, w = rnd(N, K, seed=1), rnd(N, seed=2), rnd(N, K, seed=3, scale=0.05)
    gamma, beta = 1.0 + 0.3 * rnd(K, seed=4), 0.2 * rnd(K, seed=5)
    mk = lambda: [dw.clone(), torch.empty(K, device=DEV), torch.empty(K, device=DEV)]
    g, r = mk(), mk()
    ops.unfold_norm_grads(g[0], db, w, gamma, beta, g[1], g[2])
    MockOps().unfold_norm_grads(r[0], db, w, gamma, beta, r[1], r[2])
    for n, u, v in zip(['dw', 'dgamma', 'dbeta'], g, r):
        check(f'unfold_norm_grads.{n}.N{N}.K{K}', u, v, 2e-5)


@pytest.mark.parametrize('M,N,C', [(4131, 1536, 512), (4131, 1024, 512), (1000, 192, 64)])
def test_folded_layernorm_backward_identity(ops, M, N, C):
    """The identity itself, end to end through the kernels, on CONSISTENT data: Y = bf16(xhat W'^T + b') as the forward GEMM
    stores it, row dots of a random dY with rsum and (Y - b') as the producers emit them, mbx_lnbwd_rowc, mbx_gemm_nt_lnbwd --
    against nn.LayerNorm's backward of d(xhat) = dY W' written out in fp64.  What separates the two is only the bf16 rounding
    of Y inside the second dot: the LayerNorm part of dx must agree to well below bf16 resolution (gate 1e-3)."""
    x = rnd(M, C, seed=1) * 2 + 0.3
    mu = x.mean(-1, keepdim=True)
    rs = torch.rsqrt(((x - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
    xhat = ((x - mu) * rs).to(BF)
    wf = rnd(N, C, seed=2, scale=0.15).to(BF)                       # W' [N, C]
    bf = rnd(N, seed=3, scale=0.3)
    y = (xhat.float() @ wf.float().t() + bf).to(BF)
    dy = rnd(M, N, seed=4, dtype=BF)
    rsum = wf.float().sum(1)
    nb = N // 64
    d = dy.float()
    part = torch.stack([(d * rsum).reshape(M, nb, 64).sum(-1), (d * (y.float() - bf)).reshape(M, nb, 64).sum(-1)], -1).transpose(0, 1).contiguous()
    rowc = torch.empty(M, 4, device=DEV)
    ops.lnbwd_rowc(part, rs[:, 0].contiguous(), rowc, C)
    dres = rnd(M, C, seed=5)
    dx = torch.empty(M, C, device=DEV)
    ops.gemm_nt_lnbwd(dy, wf.t().contiguous(), xhat, rowc, dres, None, dx, None)
    torch.cuda.synchronize()
    dxh = d.double() @ wf.double()
    xh = xhat.double()
    ln = rs.double() * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
    check(f'fold_identity.ln_part.M{M}.N{N}.C{C}', dx.double() - dres.double(), ln, 1e-3)
    # the leak: what the result has along the two directions nn.LayerNorm's backward projects out, relative to the result
    got = dx.double() - dres.double()
    REPORT[f'fold_identity.leak.M{M}.N{N}.C{C}'] = dict(ones=float((got.mean(-1) / rs.double()[:, 0]).abs().mean() / (ln / rs.double()).abs().mean()),
                                                        xhat=float(((got / rs.double()) * xh).mean(-1).abs().mean() / (ln / rs.double()).abs().mean()))


@pytest.mark.parametrize('name', ['lite_2x81', 'full_1x243'])
def test_folded_backward_equals_plain_backward(name):
    """The whole bf16 model, both formulations, on a reference-minted fixture (real fp64 gradients of the reference): the two
    differ in WHERE values are rounded (xhat instead of xhat g + b as the GEMM operand; an fp32 d(xhat) straight from the
    accumulator instead of a bf16 d(xn)), not in the arithmetic.  Gate: the folded path is not further from the reference than
    the plain one by more than a quarter (output, global gradient), no tensor more than 2x + 1 % of the global norm (above the 8 % floor)."""
    from tests.test_gpu_model import _fixture_grad_errors
    from tests.helpers import rel_l2, trained_like
    z, cfg = load_golden(name)
    model = build_model(cfg, seed=0)
    if int(z['trained_seed']) >= 0:
        trained_like(model, int(z['trained_seed']))
    model = model.to(DEV)
    model.precision = 'bf16'
    res = {}
    for fold in (True, False):
        model.fold_ln = fold
        model.zero_grad(set_to_none=True)
        x = torch.from_numpy(z['x']).to(DEV).requires_grad_(True)
        out = model(x)
        (out * torch.from_numpy(z['cot']).to(DEV)).sum().backward()
        e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
        res[fold] = dict(out=rel_l2(out.detach().cpu().numpy(), z['out']), dx=rel_l2(x.grad.cpu().numpy(), z['dx']), grad_global=e_all,
                         worst=e_worst, worst_name=worst, per=per)
    REPORT[f'fold_vs_plain.{name}'] = {('fold' if f else 'plain'): {k: v for k, v in r.items() if k != 'per'} for f, r in res.items()}
    top = sorted(res[True]['per'], key=lambda n: -res[True]['per'][n])[:12]
    REPORT[f'fold_vs_plain.{name}']['worst12_fold_vs_plain'] = {n: (round(res[True]['per'][n], 4), round(res[False]['per'][n], 4)) for n in top}
    f, p = res[True], res[False]
    assert f['out'] < 1.25 * p['out'] + 1e-3 and f['dx'] < 1.25 * p['dx'] + 1e-3, (f['out'], p['out'], f['dx'], p['dx'])
    assert f['grad_global'] < 1.25 * p['grad_global'] + 1e-3, (f['grad_global'], p['grad_global'])
    # per tensor: twice the plain path's error + 1 % of the global norm -- above the 8 % floor of the per-tensor bf16 gate of
    # tests/test_gpu_model.py only: on the chaotic single-clip fixture the nine level-0 tensors of the ts branch sit at 0.02-0.065
    # in EITHER formulation and move by up to 3x between rounding realisations (profiles/r03_fold_numerics.txt); after the packed
    # GELU forms changed the realisation, this run had them at 0.052 (folded) against 0.021 (plain)
    bad = {n: (f['per'][n], p['per'][n]) for n in f['per'] if f['per'][n] > max(2 * p['per'][n] + 0.01, 0.08)}
    assert not bad, bad
