"""Row-owner GEMMs (csrc/gemm_rows.hip: mbx_rows_pack_nk / mbx_rows_gemm_nk / mbx_rows_gemm_nk_ln) on a real MI355X: the qkv Linear
behind norm1 of a Block (reference lib/model/DSTformer.py:139-143 inside Block.forward :241-249).

Bars: the plain and the raw-operand form multiply the same bf16 operands in the same k order as the tile kernels and apply the same
fp32 epilogue, so the plain form must agree with mbx_gemm_nt BIT FOR BIT; against the torch restatement
(oracle/torch_ops.MockOps on the GPU; different summation order) bf16 outputs agree to 4e-3 relative L2.  The form that starts from
the fp32 rows takes the LayerNorm statistics in fp32 from those rows: against the restatement 4e-3, and its statistics path is
checked through a LayerNorm'd input with known constants."""
import pytest
import torch

from motionbert_amd.engine import EPI_RESID
from tests.mock_ops import MockOps
from tests.test_gpu_kernels import DEV, check, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPI_STORE = 0


@pytest.fixture(scope='module')
def ops():
    from motionbert_amd import hip_ops
    return hip_ops.get()


SHAPES = [(1000, 1536, 512), (128 * 5 + 17, 768, 256), (5, 64, 512), (4131, 1024, 512), (256, 512, 512), (33, 64, 256),
          (2 * 243 * 17, 1536, 512)]


def _operands(M, N, K, seed):
    a = rnd(M, K, seed=seed, dtype=BF)
    w = rnd(N, K, seed=seed + 1, dtype=BF, scale=0.05)
    bias = rnd(N, seed=seed + 2, scale=0.5)
    return a, w, bias


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_rows_gemm_store_is_the_tile_kernel_bit_for_bit(ops, M, N, K):
    a, w, bias = _operands(M, N, K, seed=M + N)
    packed = ops.rows_pack_nk(w)
    out = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    ops.rows_gemm_nk(a, packed, bias, out)
    ref = torch.empty_like(out)
    MockOps().rows_gemm_nk(a, w, bias, ref)
    check(f'rows_gemm_nk.{M}x{N}x{K}', out, ref, 4e-3)
    if N >= 256 and K % 64 == 0:      # the tile kernels' shape constraints
        tile = torch.empty_like(out)
        ops.gemm_nt(a, w, bias, EPI_STORE, out_t=tile)
        assert torch.equal(out.view(torch.int16), tile.view(torch.int16)), 'row-owner and tile kernel differ'
    # no bias: NULL is zeros
    ops.rows_gemm_nk(a, packed, None, out)
    MockOps().rows_gemm_nk(a, w, None, ref)
    check(f'rows_gemm_nk.nobias.{M}x{N}x{K}', out, ref, 4e-3)


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_rows_gemm_raw_operand_layernorm(ops, M, N, K):
    a, w, bias = _operands(M, N, K, seed=2 * M + N)
    rsum = w.float().sum(1)
    mean, rstd = rnd(M, seed=5, scale=0.2), rnd(M, seed=6).abs() + 0.5
    packed = ops.rows_pack_nk(w)
    out = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    ops.rows_gemm_nk(a, packed, bias, out, rsum, mean, rstd)
    ref = torch.empty_like(out)
    MockOps().rows_gemm_nk(a, w, bias, ref, rsum, mean, rstd)
    check(f'rows_gemm_nk.ln.{M}x{N}x{K}', out, ref, 4e-3)


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_rows_gemm_from_the_fp32_rows(ops, M, N, K):
    """Linear'(LayerNorm(x)) straight from the residual stream: rows with a mean of their own and unequal scales."""
    eps = 1e-6
    x = rnd(M, K, seed=M + 3) * (0.5 + rnd(M, 1, seed=M + 7).abs()) + 0.7 * rnd(M, 1, seed=M + 8)
    w = rnd(N, K, seed=M + 4, dtype=BF, scale=0.05)
    bias = rnd(N, seed=M + 5, scale=0.5)
    rsum = w.float().sum(1)
    packed = ops.rows_pack_nk(w)
    out = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    ops.rows_gemm_nk_ln(x, packed, bias, rsum, eps, out)
    ref = torch.empty_like(out)
    MockOps().rows_gemm_nk_ln(x, w, bias, rsum, eps, ref)
    check(f'rows_gemm_nk_ln.{M}x{N}x{K}', out, ref, 4e-3)
    # the same values through the raw-operand form with statistics computed in fp64: isolates the in-kernel statistics
    xd = x.double()
    mu = xd.mean(-1)
    rs = torch.rsqrt(((xd - mu[:, None]) ** 2).mean(-1) + eps)
    two = torch.empty_like(out)
    # (the kernel's operand is the row shifted by its first element, the statistics are those of the shifted row)
    ops.rows_gemm_nk((x - x[:, :1]).to(BF), packed, bias, two, rsum, (mu - xd[:, 0]).float(), rs.float())
    d = (out.float() - two.float()).abs()
    ulp = two.float().abs().clamp_min(2.0 ** -6) * 2.0 ** -7      # one bf16 step at the value's magnitude
    assert float((d / ulp).max()) <= 2.0, f'in-kernel LayerNorm statistics: {float((d / ulp).max()):.2f} bf16 steps'
    assert float((d > 0).float().mean()) < 0.02


@pytest.mark.parametrize('offset', [0.0, 40.0, -300.0])
def test_rows_gemm_from_rows_with_a_large_common_offset(ops, offset):
    """Against the EXACT Linear(LayerNorm(x)) (fp64 statistics and normalisation, the bf16 weights) on rows whose mean is up to 600x
    their spread -- what a trained residual stream may carry.  The kernel rounds the row shifted by its first element, so
    the error stays at the bf16 level of the normalised operand whatever the offset; rounding the raw row would lose
    sqrt(1 + mean^2 / var) of it (0.25 / 1.9 relative L2 at these offsets instead of ~3e-3)."""
    M, N, K, eps = 4131, 1536, 512, 1e-6
    x = rnd(M, K, seed=11) * 0.5 + offset
    w = rnd(N, K, seed=12, dtype=BF, scale=0.05)
    bias = rnd(N, seed=13, scale=0.5)
    out = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    ops.rows_gemm_nk_ln(x, ops.rows_pack_nk(w), bias, w.float().sum(1), eps, out)
    xd = x.double()
    mu = xd.mean(-1, keepdim=True)
    xhat = (xd - mu) * torch.rsqrt(((xd - mu) ** 2).mean(-1, keepdim=True) + eps)
    exact = xhat @ w.double().t() + bias.double()
    err = float((out.double() - exact).norm() / exact.norm())
    assert err < 8e-3, f'offset {offset}: {err:.2e} against the exact LayerNorm + Linear'


def test_rows_gemm_rejects_bad_shapes(ops):
    a, w, bias = _operands(64, 64, 128, seed=1)
    with pytest.raises(RuntimeError):
        ops.rows_pack_nk(w)                                  # K = 128
    a, w, bias = _operands(64, 96, 256, seed=1)
    with pytest.raises(RuntimeError):
        ops.rows_pack_nk(w)                                  # N % 64


LNBWD_SHAPES = [(128, 512, 512), (4131, 1536, 512), (4131, 1024, 512), (70227, 1536, 512), (33, 512, 512), (2 * 243 * 17, 1024, 512), (129, 1536, 512),
                (264384, 1024, 512), (300, 768, 512),
                # dim_feat 256 (MotionBERT-Lite, round 6): dX of qkv (K = 768) and of fc1 (K = 1024), one-trip and ragged cases
                (128, 256, 256), (4131, 768, 256), (4131, 1024, 256), (33, 512, 256), (129, 768, 256), (70227, 768, 256), (264384, 1024, 256)]


@pytest.mark.parametrize('M,K,N', LNBWD_SHAPES)
def test_rows_lnbwd_t(ops, M, K, N):
    """mbx_rows_lnbwd_t (csrc/gemm_rows_n.hip): dX GEMM of a folded (LayerNorm -> Linear) pair + LayerNorm backward with the row means
    taken from the accumulators, against the torch restatement (fp32 product of the same bf16 operands, exact means) and -- the
    identity the kernel rests on -- against LayerNorm's backward by autograd on the same dxhat."""
    dy = rnd(M, K, seed=M + 1, dtype=BF, scale=0.5)
    w = rnd(N, K, seed=M + 2, dtype=BF, scale=0.05)
    x = rnd(M, N, seed=M + 3) * (0.5 + rnd(M, 1, seed=M + 4).abs()) + 0.3 * rnd(M, 1, seed=M + 5)
    mu = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(((x - mu) ** 2).mean(-1) + 1e-6)
    xhat = ((x - mu) * rstd[:, None]).to(BF)
    dres = rnd(M, N, seed=M + 6, dtype=BF)
    out = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    ops.rows_lnbwd_t(dy, ops.rows_n_pack(w), xhat, rstd, dres, out)
    ref = torch.empty_like(out)
    MockOps().rows_lnbwd_t(dy, w, xhat, rstd, dres, ref)
    check(f'rows_lnbwd_t.{M}x{K}x{N}', out, ref, 4e-3)
    # the branch alone (dx - dres) against autograd through the plain normalisation of the same rows, with d(xhat) = dy . w^T in fp32:
    # the bf16 xhat the kernel reads differs from the exact one by its rounding, hence 1e-2
    if M <= 8192:
        xg = x.clone().requires_grad_(True)
        xh = (xg - xg.mean(-1, keepdim=True)) * torch.rsqrt(xg.var(-1, unbiased=False, keepdim=True) + 1e-6)
        xh.backward(dy.float() @ w.float().t())
        check(f'rows_lnbwd_t.vs_autograd.{M}x{K}x{N}', out.float() - dres.float(), xg.grad, 2e-2)
    assert torch.isfinite(out.float()).all()


RESID_SHAPES = [(128, 512, 512), (4131, 512, 512), (4131, 1024, 512), (70227, 512, 512), (33, 1024, 512), (2 * 243 * 17, 512, 512), (129, 1536, 512),
                (264384, 1024, 512), (300, 768, 512),
                # dim_feat 256 (MotionBERT-Lite, round 6): proj (K = 256: the single, peeled trip) and fc2 (K = 1024)
                (128, 256, 256), (4131, 256, 256), (4131, 1024, 256), (33, 512, 256), (129, 768, 256), (70227, 256, 256), (264384, 1024, 256)]


@pytest.mark.parametrize('M,K,N', RESID_SHAPES)
def test_rows_resid_ln(ops, M, K, N):
    """mbx_rows_resid_ln (csrc/gemm_rows_n.hip): y = resid + a . w^T + bias in fp32 and the plain LayerNorm of its rows (xhat, mean, rstd)
    from the same registers, against the torch restatement and against what the product ran before -- the residual epilogue of
    mbx_gemm_nt followed by mbx_layernorm_fwd; rows with a large common offset included (the statistics are two-pass)."""
    a = rnd(M, K, seed=M + 1, dtype=BF, scale=0.7)
    w = rnd(N, K, seed=M + 2, dtype=BF, scale=0.05)
    bias = rnd(N, seed=M + 3, scale=0.3)
    resid = rnd(M, N, seed=M + 4) * (0.5 + rnd(M, 1, seed=M + 5).abs()) + 3.0 * rnd(M, 1, seed=M + 6)
    mk = lambda: [torch.full((M, N), float('nan'), device=DEV), torch.full((M, N), float('nan'), device=DEV, dtype=BF),
                  torch.full((M,), float('nan'), device=DEV), torch.full((M,), float('nan'), device=DEV)]
    g, r, o = mk(), mk(), mk()
    ops.rows_resid_ln(a, ops.rows_n_pack(w), bias, resid, *g, 1e-6)
    MockOps().rows_resid_ln(a, w, bias, resid, *r, 1e-6)
    tag = f'{M}x{K}x{N}'
    for name, u, v, tol in zip(('y', 'xhat', 'mean', 'rstd'), g, r, (2e-6, 4e-3, 1e-5, 1e-5)):
        assert torch.isfinite(u.float()).all(), name
        check(f'rows_resid_ln.{name}.{tag}', u, v, tol)
    ops.gemm_nt(a, w, bias, EPI_RESID, resid=resid, out_f=o[0])
    ops.layernorm_fwd(o[0], None, None, 1e-6, o[1], o[2], o[3])
    check(f'rows_resid_ln.y_vs_tile_kernel.{tag}', g[0], o[0], 2e-6)
    check(f'rows_resid_ln.xhat_vs_layernorm_fwd.{tag}', g[1], o[1], 4e-3)
    check(f'rows_resid_ln.rstd_vs_layernorm_fwd.{tag}', g[3], o[3], 1e-5)


def test_rows_resid_ln_rejects_bad_shapes(ops):
    a, w = rnd(64, 512, seed=1, dtype=BF), rnd(512, 512, seed=2, dtype=BF)
    pk = ops.rows_n_pack(w)
    f = lambda *s: torch.empty(*s, device=DEV)
    with pytest.raises(RuntimeError):
        ops.rows_resid_ln(a, pk, f(128), f(64, 128), f(64, 128), torch.empty(64, 128, device=DEV, dtype=BF), f(64), f(64), 1e-6)      # N not 256 / 512
    with pytest.raises(RuntimeError):
        ops.rows_resid_ln(rnd(64, 384, seed=3, dtype=BF), pk, f(512), f(64, 512), f(64, 512), torch.empty(64, 512, device=DEV, dtype=BF), f(64), f(64), 1e-6)   # K % 256


def test_rows_kernels_reject_aliased_outputs(ops):
    """rows past M re-read row M - 1's inputs after its outputs were stored, so an output must not alias the input it is made
    from -- the entries check it instead of leaving it to the header's fine print."""
    a, w = rnd(64, 512, seed=1, dtype=BF), rnd(512, 512, seed=2, dtype=BF)
    pk = ops.rows_n_pack(w)
    f = lambda *s: torch.empty(*s, device=DEV)
    resid = f(64, 512)
    with pytest.raises(RuntimeError):
        ops.rows_resid_ln(a, pk, f(512), resid, resid, torch.empty(64, 512, device=DEV, dtype=BF), f(64), f(64), 1e-6)      # y is resid
    dres = rnd(64, 512, seed=3, dtype=BF)
    with pytest.raises(RuntimeError):
        ops.rows_lnbwd_t(a, pk, rnd(64, 512, seed=4, dtype=BF), f(64), dres, dres)                                        # dx_t is dres_t


def test_rows_n_pack_many_separate_allocations(ops):
    """operands that are NOT views of one flat buffer (the descriptor cache's offsets would mean nothing) still pack
    correctly -- absolute addresses in the key -- and the shared descriptor dict keeps the other packers' entries."""
    ws = [rnd(512, K, seed=10 + K, dtype=BF).clone() for K in (512, 1024)]
    ops._desc_cache[('fold', 'sentinel')] = dict()
    try:
        for _ in range(20):      # more distinct address sets than the cache keeps: only 'rnpack' entries are evicted
            many = ops.rows_n_pack_many([w.clone() for w in ws])
        assert ('fold', 'sentinel') in ops._desc_cache
        assert sum(1 for k in ops._desc_cache if k[0] == 'rnpack') <= 17
    finally:
        del ops._desc_cache[('fold', 'sentinel')]
    for w, pk in zip(ws, many):
        assert torch.equal(pk, ops.rows_n_pack(w))


def test_rows_n_pack_many(ops):
    """One launch packs operands of different contraction lengths; every image equals the one a launch of its own makes, the fragment
    layout is the documented one (fragment (kk, nt), lane (i, g): w[32 nt + i][16 kk + 8 g .. + 7]), and a second call with the same
    layout of sources (the steady state of training) reuses the cached descriptor table."""
    flat = rnd(512 * (512 + 1536 + 1024), seed=7, dtype=BF)
    ws, off = [], 0
    for K in (512, 1536, 1024):
        ws.append(flat[off:off + 512 * K].view(512, K))
        off += 512 * K
    many = ops.rows_n_pack_many(ws)
    for w, pk in zip(ws, many):
        K = w.shape[1]
        assert pk.numel() == 512 * K * 2
        assert torch.equal(pk, ops.rows_n_pack(w.clone()))
        img = pk.view(torch.bfloat16).view(K // 16, 16, 2, 32, 8)              # [kk][nt][g][i][t]
        ref = w.view(16, 32, K // 16, 2, 8).permute(2, 0, 3, 1, 4)             # w[32 nt + i][16 kk + 8 g + t]
        assert torch.equal(img, ref)
    n_cached = len(ops._desc_cache)
    again = ops.rows_n_pack_many([w for w in ws])
    assert len(ops._desc_cache) == n_cached and all(torch.equal(x, y) for x, y in zip(many, again))


def test_rows_lnbwd_t_rejects_bad_shapes(ops):
    dy, w = rnd(64, 384, seed=1, dtype=BF), rnd(512, 384, seed=2, dtype=BF)
    with pytest.raises(RuntimeError):
        ops.rows_n_pack(w)                                   # K % 256
    with pytest.raises(RuntimeError):
        ops.rows_n_pack(rnd(128, 512, seed=3, dtype=BF))     # N not 256 / 512
    with pytest.raises(RuntimeError):
        ops.rows_n_pack(rnd(512, 256, seed=4, dtype=BF))     # N = 512 needs K >= 512
