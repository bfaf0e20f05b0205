"""The no-grad path's kernels on a real MI355X against their torch restatement (oracle/torch_ops.MockOps, run on the GPU):
the fused MLP forward (mbx_mlp_fused_fwd: LayerNorm'd or raw operand -> fc1 -> GELU -> fc2 -> + residual, reference
lib/model/DSTformer.py:79-85 inside Block.forward :241-249) and the same kernel with the attention's proj + residual in front
(mbx_proj_mlp_fused_fwd).

Tolerances (relative L2): fp32 outputs of a bf16 GEMM chain 2e-5 where the operands are identical; the MLP's y goes through ONE
bf16 rounding of the hidden in both implementations (a value that lands on the other side of a rounding boundary differs by
2^-8): branch alone 1e-3, y 2e-4; bf16 outputs 4e-3; statistics 2e-5."""
import pytest
import torch

from tests.mock_ops import MockOps
from tests.test_gpu_kernels import DEV, check, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    from motionbert_amd import hip_ops
    return hip_ops.get()


def _mlp_case(ops, M, C, hidden, raw_in, want_t, want_stats, seed=0):
    eps = 1e-6
    if raw_in:      # rows with a mean of their own and unequal scales: what a residual stream looks like
        x = rnd(M, C, seed=seed + 1) * (0.5 + rnd(M, 1, seed=seed + 7).abs()) + 0.7 * rnd(M, 1, seed=seed + 8)
        a = x.to(BF)
    else:
        x = rnd(M, C, seed=seed + 1)
        a = ((x - x.mean(-1, keepdim=True)) * torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + eps)).to(BF)
    w1, w2 = rnd(hidden, C, seed=seed + 2, dtype=BF, scale=0.06), rnd(C, hidden, seed=seed + 3, dtype=BF, scale=0.04)
    b1, b2 = rnd(hidden, seed=seed + 4, scale=0.3), rnd(C, seed=seed + 5, scale=0.3)
    rsum = w1.float().sum(1)
    packed = ops.mlp_pack_weights(w1, w2)
    mk = lambda *s, dt=torch.float32: torch.full(s, float('nan'), device=DEV, dtype=dt)
    y, y2 = mk(M, C), mk(M, C)
    yt, yt2 = (mk(M, C, dt=BF), mk(M, C, dt=BF)) if want_t else (None, None)
    (mean, rstd, mean2, rstd2) = (mk(M), mk(M), mk(M), mk(M)) if want_stats else (None,) * 4
    if raw_in == 2:      # the operand is made in the kernel from the fp32 rows of the residual
        a = None
    ops.mlp_fused_fwd(a, bool(raw_in), packed, b1, b2, rsum if raw_in else None, x, y, yt, eps, mean, rstd)
    MockOps().mlp_fused_fwd(a, bool(raw_in), (w1, w2), b1, b2, rsum, x, y2, yt2, eps, mean2, rstd2)
    tag = f'C{C}.h{hidden}.M{M}.raw{int(raw_in)}'
    check(f'mlp_fused.branch.{tag}', y - x, y2 - x, 1e-3)
    check(f'mlp_fused.y.{tag}', y, y2, 2e-4)
    if want_t:
        check(f'mlp_fused.y_t.{tag}', yt, yt2, 4e-3)
    if want_stats:      # the statistics of the rows the kernel itself wrote (fp64 from its y), and those of the restatement
        yd = y.double()
        mu = yd.mean(-1)
        rs = torch.rsqrt(((yd - mu[:, None]) ** 2).mean(-1) + eps)
        scale = float(yd.std())
        assert float((mean.double() - mu).abs().max()) < 2e-6 * max(scale, 1.0), f'mlp_fused.mean.{tag}'
        check(f'mlp_fused.rstd.{tag}', rstd, rs.float(), 2e-6)
        check(f'mlp_fused.rstd_vs_restatement.{tag}', rstd, rstd2, 1e-4)
    return a, packed, b1, b2, rsum, x, y


@pytest.mark.parametrize('raw_in', [0, 1, 2])      # 2: a = None, the operand is bf16(resid) made in the kernel
@pytest.mark.parametrize('M,C,hidden', [(128, 512, 1024), (4131, 512, 1024), (70227, 512, 1024), (2754, 256, 1024), (389, 256, 128),
                                        (389, 512, 64), (1, 512, 1024), (129, 256, 1024)])
def test_mlp_fused_fwd(ops, M, C, hidden, raw_in):
    _mlp_case(ops, M, C, hidden, raw_in, True, True)


def test_mlp_fused_fwd_optional_outputs_and_inplace(ops):
    """y_t / statistics are optional; y may alias the residual input (each element is read before it is written, by the same lane)."""
    M, C, hidden = 4131, 512, 1024
    a, packed, b1, b2, rsum, x, y = _mlp_case(ops, M, C, hidden, 1, False, False, seed=20)
    x2 = x.clone()
    ops.mlp_fused_fwd(a, 1, packed, b1, b2, rsum, x2, x2, None, 1e-6, None, None)
    torch.cuda.synchronize()
    assert torch.equal(x2, y)
    # bit-for-bit repeatable (no atomics, no order-dependent reductions)
    y3 = torch.empty_like(y)
    ops.mlp_fused_fwd(a, 1, packed, b1, b2, rsum, x, y3, None, 1e-6, None, None)
    torch.cuda.synchronize()
    assert torch.equal(y3, y)


def test_mlp_fused_matches_unfused_kernels(ops):
    """Against the kernel pair it replaces (mbx_gemm_nt GELU + RESID on the same operands): same bf16 operands, same fp32
    accumulation, the GELU polynomial of the same header -- only summation order differs."""
    from motionbert_amd.engine import EPI_GELU, EPI_RESID
    M, C, hidden = 70227, 512, 1024
    a, packed, b1, b2, rsum, x, y = _mlp_case(ops, M, C, hidden, 0, False, False, seed=30)
    w1, w2 = rnd(hidden, C, seed=32, dtype=BF, scale=0.06), rnd(C, hidden, seed=33, dtype=BF, scale=0.04)
    g = torch.empty(M, hidden, device=DEV, dtype=BF)
    ops.gemm_nt(a, w1, b1, EPI_GELU, out_t=None, out2_t=g)
    y2 = torch.empty(M, C, device=DEV)
    ops.gemm_nt(g, w2, b2, EPI_RESID, out_f=y2, resid=x)
    check('mlp_fused.vs_unfused.branch', y - x, y2 - x, 1e-3)
    check('mlp_fused.vs_unfused.y', y, y2, 2e-4)


@pytest.mark.parametrize('M,C,hidden', [(1000, 512, 1024), (128 * 3 + 5, 256, 1024), (4131, 512, 1024), (77, 512, 128), (2 * 243 * 17, 256, 1024),
                                        (70227, 512, 1024), (33000, 256, 1024)])
def test_proj_mlp_fused(ops, M, C, hidden):
    """mbx_proj_mlp_fused_fwd: attention proj + residual + LayerNorm + fc1 + GELU + fc2 + residual in one kernel, against the torch
    restatement (the proj product in fp32, then the fused-MLP restatement on it)."""
    eps = 1e-6
    x = rnd(M, C, seed=1) * (0.5 + rnd(M, 1, seed=7).abs()) + 0.7 * rnd(M, 1, seed=8)
    o = rnd(M, C, seed=9, dtype=BF)
    wp = rnd(C, C, seed=10, dtype=BF, scale=0.05)
    w1, w2 = rnd(hidden, C, seed=2, dtype=BF, scale=0.06), rnd(C, hidden, seed=3, dtype=BF, scale=0.04)
    bp, b1, b2 = rnd(C, seed=11, scale=0.3), rnd(hidden, seed=4, scale=0.3), rnd(C, seed=5, scale=0.3)
    rsum = w1.float().sum(1)
    packed = ops.proj_mlp_pack_weights(wp, w1, w2)
    y = torch.full((M, C), float('nan'), device=DEV)
    y2 = torch.empty(M, C, device=DEV)
    ops.proj_mlp_fused_fwd(o, packed, bp, b1, b2, rsum, x, y, eps)
    MockOps().proj_mlp_fused_fwd(o, (wp, w1, w2), bp, b1, b2, rsum, x, y2, eps)
    tag = f'C{C}.h{hidden}.M{M}'
    check(f'proj_mlp_fused.y.{tag}', y, y2, 2e-4)
    check(f'proj_mlp_fused.branch.{tag}', y - x, y2 - x, 1e-3)
    # the two-kernel form of the same sub-layer pair: proj + residual (fp32 out), then the fused MLP from those rows
    y1, y3 = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV)
    if M >= 256:
        ops.gemm_nt(o, wp, bp, 2, out_f=y1, resid=x)
        ops.mlp_fused_fwd(None, True, ops.mlp_pack_weights(w1, w2), b1, b2, rsum, y1, y3, None, eps, None, None)
        check(f'proj_mlp_fused.vs_two_kernels.{tag}', y, y3, 2e-4)
    # in place (y aliases resid)
    xin = x.clone()
    ops.proj_mlp_fused_fwd(o, packed, bp, b1, b2, rsum, xin, xin, eps)
    assert torch.equal(xin, y)


@pytest.mark.parametrize('offset', [0.0, 40.0, -300.0])
@pytest.mark.parametrize('proj', [False, True])
def test_fused_mlp_on_rows_with_a_large_common_offset(ops, offset, proj):
    """The fused MLP (operand made in the kernel from the fp32 rows; with `proj` from the accumulators holding x + o.Wp^T + bp) against
    the EXACT sub-layer in fp64 on rows whose mean is up to 600x their spread: the operand is the row shifted by its first
    element, so the LayerNorm's cancellation happens before the bf16 rounding, not after it."""
    M, C, hidden, eps = 4131, 512, 1024, 1e-6
    x = rnd(M, C, seed=21) * 0.5 + offset
    w1, w2 = rnd(hidden, C, seed=22, dtype=BF, scale=0.06), rnd(C, hidden, seed=23, dtype=BF, scale=0.04)
    b1, b2 = rnd(hidden, seed=24, scale=0.3), rnd(C, seed=25, scale=0.3)
    rsum = w1.float().sum(1)
    y = torch.full((M, C), float('nan'), device=DEV)
    xd = x.double()
    if proj:
        o, wp, bp = rnd(M, C, seed=26, dtype=BF), rnd(C, C, seed=27, dtype=BF, scale=0.02), rnd(C, seed=28, scale=0.3)
        ops.proj_mlp_fused_fwd(o, ops.proj_mlp_pack_weights(wp, w1, w2), bp, b1, b2, rsum, x, y, eps)
        xd = xd + o.double() @ wp.double().t() + bp.double()
    else:
        ops.mlp_fused_fwd(None, True, ops.mlp_pack_weights(w1, w2), b1, b2, rsum, x, y, None, eps, None, None)
    mu = xd.mean(-1, keepdim=True)
    xhat = (xd - mu) * torch.rsqrt(((xd - mu) ** 2).mean(-1, keepdim=True) + eps)
    branch = torch.nn.functional.gelu(xhat @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    got = y.double() - xd                 # (fp32 y = x + branch: at |x| ~ 300 the sum itself is only good to 2^-24 * 300 = 2e-5)
    err = float((got - branch).norm() / branch.norm())
    assert err < 1.2e-2, f'offset {offset} proj {proj}: branch error {err:.2e} against the exact sub-layer'


@pytest.mark.parametrize('C', [512, 256])
def test_proj_mlp_fused_rows_do_not_depend_on_the_launch_size(ops, C):
    """A wave owns the same 32 rows and runs the same instruction sequence whatever M is: the first rows of a large call and a small
    call on the same data must agree bit for bit (ragged tails included)."""
    hidden, eps, Mbig = 1024, 1e-6, 40000
    x = rnd(Mbig, C, seed=31) * (0.5 + rnd(Mbig, 1, seed=32).abs()) + 0.7 * rnd(Mbig, 1, seed=33)
    o = rnd(Mbig, C, seed=34, dtype=BF)
    wp = rnd(C, C, seed=35, dtype=BF, scale=0.05)
    w1, w2 = rnd(hidden, C, seed=36, dtype=BF, scale=0.06), rnd(C, hidden, seed=37, dtype=BF, scale=0.04)
    bp, b1, b2 = rnd(C, seed=38, scale=0.3), rnd(hidden, seed=39, scale=0.3), rnd(C, seed=40, scale=0.3)
    rsum = w1.float().sum(1)
    packed = ops.proj_mlp_pack_weights(wp, w1, w2)
    big = torch.empty(Mbig, C, device=DEV)
    ops.proj_mlp_fused_fwd(o, packed, bp, b1, b2, rsum, x, big, eps)
    for M in (4131, 12000, 64, 33):
        small = torch.full((M, C), float('nan'), device=DEV)
        ops.proj_mlp_fused_fwd(o[:M].contiguous(), packed, bp, b1, b2, rsum, x[:M].contiguous(), small, eps)
        assert torch.equal(small, big[:M]), f'M = {M}: differs from the large launch'
