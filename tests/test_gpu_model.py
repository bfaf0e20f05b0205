"""End-to-end parity of the HIP DSTformer on a real MI355X.

Gate (BASELINE.json north_star): outputs within 1e-3 relative (fp32) of the reference DSTformer on
identical inputs.  The gate is asserted in `precision='fp32'` (exact fp32 MFMA).  The bf16 mode
(the throughput mode) is asserted at the bf16 noise floor the reference itself shows under
torch.autocast (7e-3 .. 4e-2) and its measured error is written to
gpurun_out/model_parity.json.
References: tests/golden/*.npz (minted from the real reference), the numpy oracle, and the torch
restatement of the kernel set (tests/mock_ops.py) run on the GPU for the shape sweep."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from motionbert_amd import model as M
from tests.helpers import build_model, load_golden, make_input, oracle_cfg, rel_l2, set_switch, trained_like
from tests.mock_ops import MockOps

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REPORT = {}
TOL_FP32 = 1e-3          # the north-star gate
TOL_BF16_OUT = 4e-2      # bf16 noise floor of the reference under autocast
TOL_BF16_GRAD = 6e-2     # global relative L2 over all parameter gradients in bf16 mode

# ---- Gate ledger: every tolerance of this file, and the commit that last changed it.  FROZEN since round 3:
# a kernel change that needs a looser number here is a kernel bug until proven otherwise; round 4 changed none of them.
#   gate                                                                     value                                   last changed
#   fp32 / bf16x3: output, dx, global and per-tensor gradient (fixtures)     TOL_FP32 = 1e-3                         5c6b953 (round 1)
#   bf16: output                                                             < min(2 x autocast, max(4e-2, autocast)) aaabbfb (round 2)
#   bf16: global gradient                                                    < min(2 x autocast, max(0.08, autocast)) aaabbfb (round 2)
#   bf16: per-tensor gradient                                                <= max(3 x autocast, 0.08 of the global norm)   def31fb (round 3: 0.05 -> 0.08
#                                                                              with LayerNorm folding, profiles/r03_fold_numerics.txt)
#   bf16: input gradient (round 6, NEW: it was recorded and never asserted) < min(2 x autocast_dx, max(0.08, autocast_dx))  round 6 (the form of the global-gradient gate)
#   bf16x3 per tensor, chaotic oracle case / recompute fixtures              3e-3                                    347c9a1 / 7ad8d74 (round 2 / 3)
#   bf16 on the chaotic oracle case                                          out < 0.0778, global gradient < 1.509   fc24144 (round 2: the
#                                                                              reference's own autocast error on that configuration)
#   TOL_BF16_GRAD (tiny fixtures, shape sweep)                               6e-2                                    7b55b57 (round 1)
#   fold vs plain per tensor (tests/test_gpu_fold.py)                        <= max(2 x plain + 0.01, 0.08)          6559556 (round 3)
#   no-grad path (round 4, new test, test_no_grad_path_on_fixtures)          the bf16 output gate above, unchanged


def grad_errors(got, ref):
    """(global rel-L2 over all tensors, worst per-tensor error, its name).  A tensor's error is
    ||got-ref|| / max(||ref||, 1% of the global gradient norm): gradients that are tiny, nearly
    cancelling sums over tokens (ts_attn.*.bias: d/db0 = -d/db1 = sum of signed per-token terms) are
    judged against the scale of the whole gradient, not against their own rounding noise."""
    names = [n for n in ref]
    g = np.sqrt(sum(float(np.sum(np.asarray(ref[n], np.float64) ** 2)) for n in names))
    d = np.sqrt(sum(float(np.sum((np.asarray(got[n], np.float64) - np.asarray(ref[n], np.float64)) ** 2)) for n in names))
    per = {n: float(np.linalg.norm(np.asarray(got[n], np.float64) - np.asarray(ref[n], np.float64)) /
                    max(np.linalg.norm(np.asarray(ref[n], np.float64)), 0.01 * g)) for n in names}
    worst = max(per, key=per.get)
    return d / g, per[worst], worst
LITE = dict(dim_in=3, dim_out=3, dim_feat=256, dim_rep=512, depth=5, num_heads=8, mlp_ratio=4, num_joints=17, maxlen=243)
FULL = dict(dim_in=3, dim_out=3, dim_feat=512, dim_rep=512, depth=5, num_heads=8, mlp_ratio=2, num_joints=17, maxlen=243)


@pytest.fixture(scope='module', autouse=True)
def _dump_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'model_parity.json'), 'w') as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def test_native_library_is_what_runs():
    from motionbert_amd import hip_ops
    ops = hip_ops.get()
    assert ops.lib.mbx_version() >= 100
    with open('/proc/self/maps') as f:
        assert 'libmbx.so' in f.read(), 'libmbx.so is not mapped into this process'


def _golden_model(name, precision):
    z, cfg = load_golden(name)
    model = build_model(cfg)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}, strict=True)
    model.precision = precision
    return z, model.to(DEV)


@pytest.mark.parametrize('name', ['tiny_default', 'tiny_trained'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
def test_tiny_golden_forward_backward(name, precision):
    z, model = _golden_model(name, precision)
    x = torch.from_numpy(z['x']).to(DEV).requires_grad_(True)
    out = model(x)
    e_out = rel_l2(out.detach().cpu().numpy(), z['out'])
    (out * torch.from_numpy(z['cot']).to(DEV)).sum().backward()
    e_dx = rel_l2(x.grad.cpu().numpy(), z['dx'])
    e_all, e_worst, worst = grad_errors({n: p.grad.cpu().numpy() for n, p in model.named_parameters()},
                                        {n: z['g.' + n] for n, _ in model.named_parameters()})
    REPORT[f'{name}.{precision}'] = dict(out=e_out, dx=e_dx, grad_global=e_all, worst_grad=e_worst, worst_name=worst)
    rep = model.get_representation(x.detach())
    e_rep = rel_l2(rep.detach().cpu().numpy(), z['rep'])
    if precision in ('fp32', 'bf16x3'):
        assert e_out < TOL_FP32 and e_rep < TOL_FP32 and e_dx < TOL_FP32, (e_out, e_rep, e_dx)
        assert e_all < TOL_FP32 and e_worst < TOL_FP32, (e_all, worst, e_worst)
    else:
        assert e_out < TOL_BF16_OUT and e_rep < TOL_BF16_OUT, (e_out, e_rep)
        assert e_all < TOL_BF16_GRAD and e_worst < 0.15, (e_all, worst, e_worst)


def test_tiny_golden_representation_path_gradients():
    z, model = _golden_model('tiny_trained', 'fp32')
    x = torch.from_numpy(z['x']).to(DEV).requires_grad_(True)
    rep = model.get_representation(x)
    (rep * torch.from_numpy(z['cot_rep']).to(DEV)).sum().backward()
    assert rel_l2(x.grad.cpu().numpy(), z['dx_rep']) < TOL_FP32
    assert model.head.weight.grad is None and model.head.bias.grad is None   # like the reference's autograd
    for n, p in model.named_parameters():
        if np.linalg.norm(z['grep.' + n]) > 1e-6:
            assert rel_l2(p.grad.cpu().numpy(), z['grep.' + n]) < TOL_FP32, n


@pytest.mark.parametrize('name', ['lite', 'full'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_seed0_models_match_reference(name, precision):
    """Model built exactly like load_backbone (learning.py:83-85) from seed 0: output and gradient
    statistics against the reference's fp64 run stored in tests/golden/seed0_*.npz."""
    z, cfg = load_golden('seed0_' + name)
    model = build_model(cfg, seed=0).to(DEV)
    model.precision = precision
    x = torch.from_numpy(z['x']).to(DEV)
    out = model(x)
    e_out = rel_l2(out.detach().cpu().numpy(), z['out'])
    (out * torch.from_numpy(z['cot']).to(DEV)).sum().backward()
    names = [str(n) for n in z['names']]
    got = {n: p.grad.double() for n, p in model.named_parameters()}
    l2 = np.asarray([got[n].norm().item() for n in names])
    ref_l2 = z['g_stats'][:, 0]
    mask = ref_l2 > 1e-9
    e_g = float(np.max(np.abs(l2[mask] - ref_l2[mask]) / ref_l2[mask]))
    e_total = abs(np.sqrt((l2 ** 2).sum()) - np.sqrt((ref_l2 ** 2).sum())) / np.sqrt((ref_l2 ** 2).sum())
    REPORT[f'seed0_{name}.{precision}'] = dict(out=e_out, worst_grad_norm=e_g, total_grad_norm=float(e_total))
    if precision == 'fp32':
        assert e_out < TOL_FP32 and e_g < TOL_FP32, (e_out, e_g)
    else:
        assert e_out < TOL_BF16_OUT and e_total < 5e-2, (e_out, e_total)


def _fixture_grad_errors(model, z):
    """Errors of the parameter gradients against a baseline-shape fixture (oracle/make_golden.py baseline_shape):
    (global rel-L2 estimated from the stored elements, worst per-tensor error, its name, worst norm mismatch).
    Small tensors are stored in full, large ones as a fixed 4096-element sample; a sampled tensor's error norm is
    scaled by sqrt(numel / sample) and, like everywhere in this file, a tensor is judged against
    max(its own reference norm, 1 % of the global gradient norm)."""
    names = [str(n) for n in z['names']]
    ref_l2 = z['g_stats'][:, 0]
    g_glob = float(np.sqrt((ref_l2 ** 2).sum()))
    params = dict(model.named_parameters())
    d2, per, norm_err = 0.0, {}, 0.0
    for k, n in enumerate(names):
        got = params[n].grad.detach().double().reshape(-1).cpu().numpy()
        if 'g.' + n in z.files:
            ref, scale = z['g.' + n].astype(np.float64), 1.0
            d = got - ref
        else:
            idx = z[f'idx.{got.size}']
            ref, scale = z['gs.' + n].astype(np.float64), got.size / len(idx)
            d = got[idx] - ref
        e2 = float((d ** 2).sum()) * scale
        d2 += e2
        per[n] = np.sqrt(e2) / max(ref_l2[k], 0.01 * g_glob)
        norm_err = max(norm_err, abs(float(np.linalg.norm(got)) - ref_l2[k]) / max(ref_l2[k], 0.01 * g_glob))
    worst = max(per, key=per.get)
    return np.sqrt(d2) / g_glob, per[worst], worst, norm_err, per


@pytest.mark.parametrize('recompute', [False, True])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('name', ['lite_2x81', 'full_1x243'])
def test_baseline_shape_fixture_fwd_bwd(name, precision, recompute):
    """Reference-minted fixtures at the BASELINE.json shapes: MotionBERT-Lite on [2,81,17,3]
    (configs[0]) and the full model on [1,243,17,3] -- output, input gradient and REAL parameter gradients (full for
    small tensors, a fixed sample of every large one, the norm of all 260) of the reference's own fp64 autograd.
    fp32 mode: the 1e-3 gate, per tensor.  bf16 mode: gated against what the reference ITSELF does under
    torch.autocast(bfloat16) on the same weights and input (numbers minted into the fixture), times 2.
    `recompute=True` is the low-memory mode (LayerNorm outputs and MLP post-activations rebuilt in backward: the path
    behind the bench line's `full_model_b256`) -- same gates."""
    z, cfg = load_golden(name)
    model = build_model(cfg, seed=0)
    if int(z['trained_seed']) >= 0:
        trained_like(model, int(z['trained_seed']))
    got_w = np.asarray([[v.double().sum().item(), v.double().abs().sum().item()] for v in model.state_dict().values()])
    # (the same seed gives the same weights up to the last ulp of erfinv on a different host CPU: 1e-9 on these sums)
    assert np.allclose(got_w, z['w_stats'], rtol=1e-6, atol=1e-6), 'weights were not re-created from the seed'
    model = model.to(DEV)
    model.precision = precision
    model.recompute = recompute
    x = torch.from_numpy(z['x']).to(DEV).requires_grad_(True)
    out = model(x)
    e_out = rel_l2(out.detach().cpu().numpy(), z['out'])
    (out * torch.from_numpy(z['cot']).to(DEV)).sum().backward()
    e_dx = rel_l2(x.grad.cpu().numpy(), z['dx'])
    e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
    ac = dict(out=float(z['autocast_out']), grad_global=float(z['autocast_grad_global']), worst_grad=float(z['autocast_grad_per'].max()))
    REPORT[f'fixture.{name}.{precision}' + ('.recompute' if recompute else '')] = dict(out=e_out, dx=e_dx, grad_global=e_all, worst_grad=e_worst, worst_name=worst,
                                                 worst_norm_mismatch=e_norm, reference_autocast_bf16=ac)
    if precision in ('fp32', 'bf16x3'):     # the two modes that carry the north-star 1e-3 gate
        assert e_out < TOL_FP32 and e_dx < TOL_FP32, (e_out, e_dx)
        assert e_all < TOL_FP32 and e_worst < TOL_FP32 and e_norm < TOL_FP32, (e_all, worst, e_worst, e_norm)
    else:
        # never looser than twice what the reference shows under autocast on this very fixture, and never above the fixed
        # bf16 bounds (output 4e-2; global gradient 0.08) unless the reference's own autocast error is above them (the
        # trained-like single-clip fixture: reference 0.22 on the global gradient, this path 0.11 -- measured round 2)
        assert e_out < min(2 * ac['out'], max(TOL_BF16_OUT, ac['out'])), (e_out, ac)
        assert e_all < min(2 * ac['grad_global'], max(0.08, ac['grad_global'])), (e_all, ac)
        names = [str(n) for n in z['names']]
        # per tensor: two bf16 pipelines are two different realisations of the same rounding noise -- measured round 2:
        # this path is 2x BETTER than the reference-under-autocast on the global gradient of full_1x243 (0.107 vs 0.217)
        # and on the late levels, and 2.2-2.5x worse on nine level-0 tensors (0.040 vs 0.017 of the global norm)
        # Round 3 (LayerNorm folding, the default bf16 backward): output, dx and the global gradient error halve or better
        # (full_1x243: 0.0169 / 0.094 / 0.018 against 0.027 / 0.139 / 0.108 in round 2) and the level-0 tensors even out -- the
        # plain backward had 0.10 on blocks_st.0 and 0.01-0.03 on blocks_ts.0, the folded one 0.02-0.05 and 0.05-0.065, one
        # rounding realisation differing from the next by up to 3x there (profiles/r03_fold_numerics.txt, four seeds).  The floor
        # of this per-tensor gate, whose job is to catch a WRONG tensor (error of order one), moves from 5 % to 8 % of the
        # global norm; the relative part (3x the reference's own autocast error) stays.
        bad = {n: (per[n], float(a)) for n, a in zip(names, z['autocast_grad_per']) if per[n] > max(3 * float(a), 0.08)}
        assert not bad, f'bf16 per-tensor gradient error above max(3x the reference-under-autocast error, 8 % of the global norm): {bad}'
        # Round 6: the INPUT gradient gets the gate every other bf16 quantity has -- against the reference's own
        # autocast error on dx, minted into the fixture by oracle/make_golden.py (autocast_dx; lite_2x81 0.0085, full_1x243 0.156).
        ac_dx = float(z['autocast_dx'])
        assert e_dx < min(2 * ac_dx, max(0.08, ac_dx)), (e_dx, ac_dx)
        # measured value / gate for every frozen bf16 gate (profiles/r06_bf16_headroom.txt is printed from these): the next rounding
        # that is "inside the spread" has a number to be inside of
        worst_ratio, worst_t = max((per[n] / max(3 * float(a), 0.08), n) for n, a in zip(names, z['autocast_grad_per']))
        REPORT[f'headroom.{name}.bf16' + ('.recompute' if recompute else '')] = dict(
            out=(e_out, min(2 * ac['out'], max(TOL_BF16_OUT, ac['out']))), dx=(e_dx, min(2 * ac_dx, max(0.08, ac_dx))),
            grad_global=(e_all, min(2 * ac['grad_global'], max(0.08, ac['grad_global']))),
            per_tensor_worst=(per[worst_t], max(3 * float(dict(zip(names, z['autocast_grad_per']))[worst_t]), 0.08), worst_t), per_tensor_worst_ratio=worst_ratio)


@pytest.mark.parametrize('name', ['tiny_trained', 'lite_2x81', 'full_1x243'])
def test_no_grad_path_on_fixtures(name, monkeypatch):
    """The inference sequencing (raw-operand LayerNorm + fused MLP, engine.py `rawln`) against the reference-minted fixtures: the
    same output gate as the training forward of the same precision, and agreement with the training sequencing run without saves
    (MBX_RAWLN=0) at the bf16 noise floor.  fp32-class modes have no separate no-grad sequencing (checked: same bits)."""
    z, cfg = load_golden(name)
    if any(k.startswith('w.') for k in z.files):      # the tiny fixtures carry their weights
        model = build_model(cfg)
        model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}, strict=True)
    else:                                             # the BASELINE-shape fixtures are rebuilt from the seed (checked against w_stats)
        model = build_model(cfg, seed=0)
        if int(z['trained_seed']) >= 0:
            trained_like(model, int(z['trained_seed']))
        got_w = np.asarray([[v.double().sum().item(), v.double().abs().sum().item()] for v in model.state_dict().values()])
        assert np.allclose(got_w, z['w_stats'], rtol=1e-6, atol=1e-6)
    model = model.to(DEV).eval()
    x = torch.from_numpy(z['x']).to(DEV)
    ac_out = float(z['autocast_out']) if 'autocast_out' in z.files else TOL_BF16_OUT
    outs = {}
    for raw in ('1', '0'):
        set_switch(monkeypatch, 'MBX_RAWLN', raw)
        with torch.no_grad():
            model.precision = 'bf16'
            outs[raw] = model(x).float().cpu().numpy()
            rep = model.get_representation(x).float().cpu().numpy()
        e_out, e_rep = rel_l2(outs[raw], z['out']), rel_l2(rep, z['rep']) if 'rep' in z.files else 0.0
        REPORT[f'nograd.{name}.bf16.rawln{raw}'] = dict(out=e_out, rep=e_rep)
        assert e_out < min(2 * ac_out, max(TOL_BF16_OUT, ac_out)), (raw, e_out, ac_out)
    assert rel_l2(outs['1'], outs['0']) < max(TOL_BF16_OUT, ac_out)
    set_switch(monkeypatch, 'MBX_RAWLN', '1')
    with torch.no_grad():
        model.precision = 'fp32'
        o32 = model(x)
    assert rel_l2(o32.cpu().numpy(), z['out']) < TOL_FP32


@pytest.mark.timeout(900)
def test_oracle_full_t243_fwd_bwd():
    """The numpy fp64 oracle itself (forward AND hand-written backward) on the full model at [1,243,17,3] with
    trained-like weights other than the fixture's: the T=243 check no longer rests on MockOps-through-the-same-engine
    (VERDICT r1 item 1b).  About a minute of CPU time on the GPU box."""
    from oracle import dstformer_oracle as O
    model = build_model(FULL, seed=7)
    trained_like(model, 8)
    P = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    x = make_input(1, 243, 17, 9)
    cot = torch.randn(1, 243, 17, 3, generator=torch.Generator().manual_seed(10))
    ref, cache = O.forward(P, x.numpy(), oracle_cfg(FULL), want_cache=True)
    G, dx = O.backward(P, cache, cot.numpy(), oracle_cfg(FULL))
    del cache
    model = model.to(DEV)
    for precision in ('fp32', 'bf16x3', 'bf16'):
        model.precision = precision
        model.zero_grad(set_to_none=True)
        xd = x.to(DEV).requires_grad_(True)
        out = model(xd)
        (out * cot.to(DEV)).sum().backward()
        e_out = rel_l2(out.detach().cpu().numpy(), ref)
        e_dx = rel_l2(xd.grad.cpu().numpy(), dx)
        e_all, e_worst, worst = grad_errors({n: p.grad.cpu().numpy() for n, p in model.named_parameters()}, G)
        REPORT[f'oracle_full_1x243.{precision}'] = dict(out=e_out, dx=e_dx, grad_global=e_all, worst_grad=e_worst, worst_name=worst)
        if precision in ('fp32', 'bf16x3'):
            # The 1e-3 gate holds for the output, dx and the global gradient in both modes.  Per tensor, this deliberately
            # chaotic configuration (3x weights, one clip) amplifies rounding by ~2000x: plain fp32 already sits at 1.1e-4 on
            # the last block's tensors, and bf16x3 (2^-17 operands instead of 2^-24) at a uniform 16x of that, 1.6-1.8e-3
            # (measured round 2, tools/x3_oracle_diag.py; deterministic run to run).  On the reference-minted fixtures the
            # same mode passes 1e-3 per tensor (test_baseline_shape_fixture_fwd_bwd).
            assert max(e_out, e_dx, e_all) < TOL_FP32, (precision, e_out, e_dx, e_all)
            assert e_worst < (TOL_FP32 if precision == 'fp32' else 3e-3), (precision, worst, e_worst)
        else:
            # bf16 yardstick for THIS configuration, minted with oracle/autocast_yardstick.py: the reference itself under
            # torch.autocast(bfloat16) is off by 0.0778 (output) and 1.509 (global gradient) against its fp64 run -- 3x
            # amplified weights and a single clip make the network chaotic at bf16 resolution.  Gate: no worse than that.
            assert e_out < 0.0778 and e_all < 1.509, (e_out, e_all)


def test_config0_lite_forward_vs_oracle():
    """BASELINE.json configs[0]: MotionBERT-Lite forward on random [2,81,17,3]; the HIP path (fp32 mode)
    against the numpy fp64 oracle on the same weights and input."""
    from oracle import dstformer_oracle as O
    model = build_model(LITE, seed=0)
    trained_like(model, 5)
    P = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    x = make_input(2, 81, 17, 3)
    ref = O.forward(P, x.numpy(), oracle_cfg(LITE))
    model = model.to(DEV).eval()
    for precision, tol in (('fp32', TOL_FP32), ('bf16', TOL_BF16_OUT)):
        model.precision = precision
        with torch.no_grad():
            out = model(x.to(DEV))
        e = rel_l2(out.cpu().numpy(), ref)
        REPORT[f'config0_lite_2x81.{precision}'] = e
        assert e < tol, (precision, e)


def _mock_reference(model, x, cot, return_rep=False, precision='fp32'):
    """Same weights through the torch restatement of the kernel set on the GPU, in fp32 or with the
    same bf16 rounding points as the HIP path."""
    saved = model.precision
    model.precision = precision
    for p in model.parameters():
        p.grad = None
    out = M.run(MockOps(), model, x, return_rep)
    (out * cot).sum().backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    model.precision = saved
    return out.detach(), grads


SWEEP = [('lite', 1, 1), ('lite', 2, 16), ('lite', 2, 30), ('lite', 1, 100), ('full', 1, 81), ('full', 2, 243), ('lite', 1, 243), ('full', 3, 33)]


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
@pytest.mark.parametrize('size,B,T', SWEEP)
def test_shape_sweep_fwd_bwd(size, B, T, precision):
    """T in {1,16,30,33,81,100,243} (every length the reference feeds the model), both sizes."""
    cfg = LITE if size == 'lite' else FULL
    model = build_model(cfg, seed=1)
    trained_like(model, 2)
    model = model.to(DEV)
    x = make_input(B, T, 17, 10 + T).to(DEV)
    cot = torch.randn(B, T, 17, 3, generator=torch.Generator().manual_seed(T)).to(DEV)
    ref, gref = _mock_reference(model, x, cot)
    model.precision = precision
    out = model(x)
    (out * cot).sum().backward()
    got = {n: p.grad.cpu().numpy() for n, p in model.named_parameters()}
    e_out = rel_l2(out.detach().cpu().numpy(), ref.cpu().numpy())
    e_all, e_worst, worst = grad_errors(got, {n: g.cpu().numpy() for n, g in gref.items()})
    rec = dict(out=e_out, grad_global=e_all, worst_grad=e_worst, worst_name=worst)
    if precision in ('fp32', 'bf16x3'):
        REPORT[f'sweep.{size}.B{B}T{T}.{precision}'] = rec
        assert e_out < TOL_FP32 and e_all < TOL_FP32 and e_worst < TOL_FP32, (e_out, e_all, worst, e_worst)
    else:
        # bf16 mode vs the fp32 run of the same weights: this is the bf16 noise floor of a 5-level network with
        # "trained-like" (3x) weights -- the reference itself under torch.autocast shows 2e-2..4e-2 on outputs
        # here.  Two different bf16 pipelines (these kernels vs a torch restatement
        # with the same rounding points) measured just as far apart (1.2e-2 / 5e-2 at [2,243]), so the bound is
        # a noise-floor bound, not a parity gate; the gate lives in the fp32 branch above.  Tiny inputs
        # (17..1700 tokens) average over too few elements and get a looser bound.
        REPORT[f'sweep.{size}.B{B}T{T}.{precision}'] = rec
        small = B * T * 17 < 2000
        assert e_out < (8e-2 if small else 5e-2) and e_all < (0.35 if small else 0.12), (e_out, e_all, worst, e_worst)


def test_infer_wild_call_pattern():
    """infer_wild.py:28-40,66-88: DataParallel wrap, strict load of 'module.'-prefixed keys, eval, B=1,
    T <= 243 tail clip, flip-TTA (two forwards), in-place write into the output."""
    model = build_model(LITE, seed=0)
    sd = {'module.' + k: v.clone() for k, v in model.state_dict().items()}
    wrapped = nn.DataParallel(build_model(LITE, seed=3)).to(DEV)
    wrapped.load_state_dict(sd, strict=True)
    wrapped.eval()
    x = make_input(1, 57, 17, 4).to(DEV)
    with torch.no_grad():
        a = wrapped(x)
        xf = x.clone()
        xf[..., 0] *= -1
        b = wrapped(xf)
        out = (a + b) / 2
        out[:, :, 0, :] = 0
    assert out.shape == (1, 57, 17, 3) and torch.isfinite(out).all()
    model = model.to(DEV).eval()
    with torch.no_grad():
        assert torch.allclose(model(x), a, atol=0, rtol=0)  # deterministic, same weights


def test_actionnet_style_head_trains_through_get_representation():
    """model_action.py:62-70 restated: [N,M,T,17,3] -> backbone.get_representation -> mean over T, M -> fc."""
    backbone = build_model(LITE, seed=0).to(DEV)
    fc = nn.Linear(17 * 512, 60).to(DEV)
    N, Mp, T = 2, 2, 27
    x = make_input(N * Mp, T, 17, 6).to(DEV)
    opt = torch.optim.AdamW(list(backbone.parameters()) + list(fc.parameters()), lr=1e-4)
    losses = []
    for _ in range(3):
        feat = backbone.get_representation(x).reshape(N, Mp, T, 17, 512)
        logits = fc(feat.mean(2).reshape(N, Mp, -1).mean(1))
        loss = nn.functional.cross_entropy(logits, torch.tensor([3, 7], device=DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert backbone.head.weight.grad is None  # unused on this path: no gradient, as in the reference


def test_non_contiguous_and_no_conf_input():
    model = build_model(LITE, seed=0).to(DEV).eval()
    x = make_input(2, 30, 17, 5).to(DEV)
    xs = torch.cat([x, x], -1)[..., :3]       # non-contiguous view
    with torch.no_grad():
        assert torch.equal(model(xs), model(x))


def test_hipgraph_replay_matches_eager():
    """B=1 clip-at-a-time inference (infer_wild.py:66-88) through a captured hipGraph."""
    import time
    from motionbert_amd.graph import GraphedForward
    model = build_model(FULL, seed=0).to(DEV).eval()
    x = make_input(1, 243, 17, 8).to(DEV)
    fast = GraphedForward(model, x)
    x2 = make_input(1, 243, 17, 9).to(DEV)
    with torch.no_grad():
        ref = model(x2)
    got = fast(x2)
    assert torch.equal(got, ref)
    with torch.no_grad():   # weights updated in place are picked up without re-capture
        model.head.bias.add_(1.0)
        assert torch.equal(fast(x2), model(x2))

    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        t_eager, t_graph = timeit(lambda: model(x2)), timeit(lambda: fast(x2))
    # report only: wall-clock numbers never gate the parity suite
    REPORT['hipgraph.B1T243'] = dict(eager_ms=t_eager, graph_ms=t_graph)


GENERIC = [
    dict(dim_in=2, dim_out=3, dim_feat=128, dim_rep=64, depth=1, num_heads=4, mlp_ratio=2, num_joints=13, maxlen=40),
    dict(dim_in=3, dim_out=5, dim_feat=64, dim_rep=128, depth=2, num_heads=2, mlp_ratio=4, num_joints=25, maxlen=50, qkv_bias=False),
    dict(dim_in=3, dim_out=3, dim_feat=256, dim_rep=512, depth=1, num_heads=4, mlp_ratio=1, num_joints=17, maxlen=243, att_fuse=False),
    dict(dim_in=4, dim_out=2, dim_feat=128, dim_rep=256, depth=1, num_heads=2, mlp_ratio=2, num_joints=32, maxlen=64, qk_scale=0.2),
]


def _oracle_reference(cfg, model, x, cot):
    """Output, parameter gradients and input gradient of the numpy fp64 oracle (forward + hand-written backward) for the
    weights of `model` (CPU) -- independent of engine.py, unlike `_mock_reference`."""
    from oracle import dstformer_oracle as O
    ocfg = oracle_cfg({k: v for k, v in cfg.items() if k not in ('qkv_bias',)})
    P = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    ref, cache = O.forward(P, x.numpy(), ocfg, want_cache=True)
    G, dx = O.backward(P, cache, cot.numpy(), ocfg)
    return ref, G, dx


ORACLE_SWEEP = [('lite', 2, 1), ('lite', 2, 16), ('lite', 2, 30), ('lite', 1, 33), ('lite', 1, 100), ('full', 1, 27)]


@pytest.mark.parametrize('size,B,T', ORACLE_SWEEP)
def test_shape_sweep_vs_numpy_oracle(size, B, T):
    """The short sequence lengths of the sweep above (T in {1,16,30,33,100}: one-wave, shared-tile and 16-wave temporal
    kernels, ragged last tiles) against the NUMPY ORACLE in the two 1e-3-class modes -- `test_shape_sweep_fwd_bwd` compares
    with MockOps through the same engine.py, where a shared sequencing bug would be invisible."""
    cfg = LITE if size == 'lite' else FULL
    model = build_model(cfg, seed=21)
    trained_like(model, 22)
    x = make_input(B, T, 17, 23 + T)
    cot = torch.randn(B, T, 17, 3, generator=torch.Generator().manual_seed(24 + T))
    ref, G, dx = _oracle_reference(cfg, model, x, cot)
    model = model.to(DEV)
    for precision in ('fp32', 'bf16x3'):
        model.precision = precision
        model.zero_grad(set_to_none=True)
        xd = x.to(DEV).requires_grad_(True)
        out = model(xd)
        (out * cot.to(DEV)).sum().backward()
        e_out = rel_l2(out.detach().cpu().numpy(), ref)
        e_dx = rel_l2(xd.grad.cpu().numpy(), dx)
        e_all, e_worst, worst = grad_errors({n: p.grad.cpu().numpy() for n, p in model.named_parameters()}, G)
        REPORT[f'oracle_sweep.{size}.B{B}T{T}.{precision}'] = dict(out=e_out, dx=e_dx, grad_global=e_all, worst_grad=e_worst, worst_name=worst)
        assert max(e_out, e_dx, e_all) < TOL_FP32, (precision, e_out, e_dx, e_all)
        # per tensor: 1e-3 in fp32; bf16x3 carries 2^-17 operands and gets the 3e-3 of test_oracle_full_t243_fwd_bwd on the
        # 3x-weight ("trained-like") models, where it measured 1.6-1.8e-3 (parity statement)
        assert e_worst < (TOL_FP32 if precision == 'fp32' else 3e-3), (precision, worst, e_worst)


@pytest.mark.parametrize('idx', range(len(GENERIC)))
def test_generic_constructor_arguments(idx):
    """Constructor arguments beyond the two shipped model sizes: other joint counts, input/output widths, head
    dims 32/64, no qkv bias, plain-average fusion (att_fuse=False, DSTformer.py:351), explicit qk_scale -- against the
    numpy fp64 oracle (forward and hand-written backward), not against MockOps through the same engine."""
    cfg = GENERIC[idx]
    model = build_model(cfg, seed=3)
    trained_like(model, 4)
    B, T, J = 3, min(37, cfg['maxlen']), cfg['num_joints']
    g = torch.Generator().manual_seed(idx)
    x = torch.rand(B, T, J, cfg['dim_in'], generator=g) * 2 - 1
    cot = torch.randn(B, T, J, cfg['dim_out'], generator=g)
    ref, G, dx = _oracle_reference(cfg, model, x, cot)
    model = model.to(DEV)
    model.precision = 'fp32'
    xd = x.to(DEV).requires_grad_(True)
    out = model(xd)
    (out * cot.to(DEV)).sum().backward()
    e_out = rel_l2(out.detach().cpu().numpy(), ref)
    e_dx = rel_l2(xd.grad.cpu().numpy(), dx)
    e_all, e_worst, worst = grad_errors({n: p.grad.cpu().numpy() for n, p in model.named_parameters()}, G)
    REPORT[f'generic.{idx}'] = dict(out=e_out, dx=e_dx, grad_global=e_all, worst_grad=e_worst, worst_name=worst)
    assert e_out < TOL_FP32 and e_dx < TOL_FP32 and e_all < TOL_FP32 and e_worst < TOL_FP32, (e_out, e_dx, e_all, worst, e_worst)
    model.precision = 'bf16'
    model.zero_grad()
    out16 = model(x.to(DEV))
    (out16 * cot.to(DEV)).sum().backward()
    assert rel_l2(out16.detach().cpu().numpy(), ref) < 8e-2


def test_recompute_mode_matches_normal_mode():
    """Low-memory mode against the normal mode on the same weights and input: fp32 / bf16x3 rebuild the
    LayerNorm output and gelu(u) from fp32 tensors with the formulas of the forward kernels -> the gradients agree to
    rounding; bf16 rebuilds the post-activation as gelu(bf16(u)) where forward applied GELU to the fp32 accumulator, so
    fc2's weight gradient sees a g that differs by up to one bf16 ulp per element -- bounded here at 1/4 of the bf16 noise
    floor of the gradients (TOL_BF16_GRAD), outputs are bit-identical in every mode (forward is the same code)."""
    model = build_model(FULL, seed=31)
    trained_like(model, 32)
    model = model.to(DEV)
    x = make_input(2, 81, 17, 33).to(DEV)
    cot = torch.randn(2, 81, 17, 3, generator=torch.Generator().manual_seed(34)).to(DEV)
    model.gelu_d = False      # (round 5) the low-memory mode keeps the pre-activation: compare it with the normal mode that does, too
    for precision, tol in (('fp32', 1e-5), ('bf16x3', 1e-5), ('bf16', TOL_BF16_GRAD / 4)):
        model.precision = precision
        res = []
        for rc in (False, True):
            model.recompute = rc
            model.zero_grad(set_to_none=True)
            out = model(x)
            (out * cot).sum().backward()
            res.append((out.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()])))
        assert torch.equal(res[0][0], res[1][0])
        rel = float((res[0][1] - res[1][1]).norm() / res[0][1].norm())
        REPORT[f'recompute_vs_normal.{precision}'] = rel
        assert rel < tol, (precision, rel)
    model.recompute = False


def test_saved_gelu_derivative_against_saved_preactivation():
    """bf16 training saves gelu'(u) from the fp32 accumulator instead of u (model.gelu_d, round 5).  Against the pre-activation mode on
    the same weights and input: the two forward epilogues differ in the erf approximation only, so the outputs agree to the bf16 mode's
    own noise; both gradients are then measured against the SAME fp32 run -- the new mode must not be further from it than the gate
    that bounds the bf16 mode itself, nor noticeably further than the old one."""
    model = build_model(FULL, seed=41)
    trained_like(model, 42)
    model = model.to(DEV)
    x = make_input(2, 81, 17, 43).to(DEV)
    cot = torch.randn(2, 81, 17, 3, generator=torch.Generator().manual_seed(44)).to(DEV)
    res = {}
    for tag, precision, gd in (('fp32', 'fp32', True), ('d', 'bf16', True), ('u', 'bf16', False)):
        model.precision, model.gelu_d = precision, gd
        model.zero_grad(set_to_none=True)
        out = model(x)
        (out * cot).sum().backward()
        res[tag] = (out.detach().clone(), torch.cat([p.grad.flatten() for p in model.parameters()]))
    model.gelu_d = True
    rel = lambda a, b: float((a - b).norm() / b.norm())
    eo_d, eo_u = rel(res['d'][0], res['fp32'][0]), rel(res['u'][0], res['fp32'][0])
    eg_d, eg_u = rel(res['d'][1], res['fp32'][1]), rel(res['u'][1], res['fp32'][1])
    REPORT['gelu_d.out_err'], REPORT['gelu_u.out_err'], REPORT['gelu_d.grad_err'], REPORT['gelu_u.grad_err'] = eo_d, eo_u, eg_d, eg_u
    assert eg_d < TOL_BF16_GRAD and eg_u < TOL_BF16_GRAD, (eg_d, eg_u)
    assert eg_d < 1.25 * eg_u + 1e-3 and eo_d < 1.25 * eo_u + 1e-3, (eo_d, eo_u, eg_d, eg_u)


@pytest.mark.parametrize('recompute', [False, True])
def test_full_size_properties(recompute):
    """BASELINE.json full size ([64,243,17,3], full model, bf16) through size-independent properties:
    determinism, independence of the clips of a batch, linearity of backward in the cotangent, additivity of the
    parameter gradients over a split of the batch (what data parallelism relies on), zero cotangent -> zero grads."""
    model = build_model(FULL, seed=0)
    trained_like(model, 1)
    model = model.to(DEV)
    model.precision = 'bf16'
    model.recompute = recompute
    B, T = 64, 243
    x = make_input(B, T, 17, 77).to(DEV)
    cot = torch.randn(B, T, 17, 3, generator=torch.Generator().manual_seed(78)).to(DEV)

    def fwd_bwd(xx, cc, scale=1.0):
        model.zero_grad(set_to_none=True)
        out = model(xx)
        (out * (cc * scale)).sum().backward()
        return out.detach(), torch.cat([p.grad.flatten() for p in model.parameters()])
    out1, g1 = fwd_bwd(x, cot)
    out2, g2 = fwd_bwd(x, cot)
    assert torch.equal(out1, out2) and torch.equal(g1, g2), 'the path must be deterministic (no atomics anywhere)'
    assert torch.isfinite(out1).all() and torch.isfinite(g1).all()
    # clips are independent: a sub-batch gives the same rows (same kernels, same per-token arithmetic) -- within the no-grad
    # sequencing (round 4: raw-operand LayerNorm + fused MLP, a different kernel set from the training forward) ...
    with torch.no_grad():
        sub = model(x[5:9])
        full_ng = model(x)
    assert torch.equal(sub, full_ng[5:9])
    # ... and within the training sequencing
    sub_g = model(x[5:9].clone().requires_grad_(True)).detach()
    assert torch.equal(sub_g, out1[5:9])
    # the two sequencings are two bf16 realisations of the same function
    assert float((full_ng - out1).norm() / out1.norm()) < TOL_BF16_OUT
    REPORT['full_size.nograd_vs_train_forward' + ('.recompute' if recompute else '')] = float((full_ng - out1).norm() / out1.norm())
    # backward is linear in the cotangent (power-of-two scale: exact in floating point up to the bf16 roundings of
    # intermediate gradients, which scale exactly too)
    _, g4 = fwd_bwd(x, cot, scale=4.0)
    assert torch.allclose(g4, 4.0 * g1, rtol=1e-5, atol=0)
    # gradients add over a batch split (sum-reduced loss): different token counts -> different dW split counts
    _, ga = fwd_bwd(x[:40], cot[:40])
    _, gb = fwd_bwd(x[40:], cot[40:])
    rel = float((ga + gb - g1).norm() / g1.norm())
    REPORT['full_size.split_additivity' + ('.recompute' if recompute else '')] = rel
    assert rel < 2e-3, rel
    _, g0 = fwd_bwd(x, torch.zeros_like(cot))
    assert float(g0.abs().max()) == 0.0


def test_empty_batch_and_max_length():
    model = build_model(LITE, seed=0).to(DEV)
    out = model(torch.zeros(0, 243, 17, 3, device=DEV))
    assert out.shape == (0, 243, 17, 3)
    out.sum().backward()
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0 for p in model.parameters())
    assert model.get_representation(torch.zeros(0, 5, 17, 3, device=DEV)).shape == (0, 5, 17, 512)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 244, 17, 3, device=DEV))   # longer than the learned temporal embedding (maxlen)


def test_get_and_reset_classifier():
    """DSTformer.get_classifier / reset_classifier (DSTformer.py:322-327): the new head is Linear(dim_feat, dim_out) -- usable
    whenever dim_feat == dim_rep, as in the reference -- and dim_out = 0 makes the head an Identity (forward returns the
    representation).  Forward + backward through the replaced head against the numpy oracle."""
    from oracle import dstformer_oracle as O
    cfg = dict(dim_in=3, dim_out=3, dim_feat=64, dim_rep=64, depth=2, num_heads=2, mlp_ratio=2, num_joints=17, maxlen=16)
    model = build_model(cfg, seed=11)
    trained_like(model, 12)
    assert model.get_classifier() is model.head
    torch.manual_seed(13)
    model.reset_classifier(5)
    assert model.get_classifier() is model.head and model.head.out_features == 5 and model.dim_out == 5
    P = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    x = make_input(2, 9, 17, 14)
    cot = torch.randn(2, 9, 17, 5, generator=torch.Generator().manual_seed(15))
    ocfg = oracle_cfg(dict(cfg, dim_out=5))
    ref, cache = O.forward(P, x.numpy(), ocfg, want_cache=True)
    G, _ = O.backward(P, cache, cot.numpy(), ocfg)
    model = model.to(DEV)
    model.precision = 'fp32'
    out = model(x.to(DEV))
    assert out.shape == (2, 9, 17, 5) and rel_l2(out.detach().cpu().numpy(), ref) < TOL_FP32
    (out * cot.to(DEV)).sum().backward()
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), G[n]) < TOL_FP32 or np.linalg.norm(G[n]) < 1e-9, n
    model.reset_classifier(0)               # Identity head: the model returns its representation
    with torch.no_grad():
        rep = model(x.to(DEV))
    assert rep.shape == (2, 9, 17, 64) and torch.equal(rep, model.get_representation(x.to(DEV)))
    model.reset_classifier(9)                # the skinny head kernel handles dim_out <= 8: a loud error, not garbage
    with pytest.raises(RuntimeError, match='dim_out'):
        model.to(DEV)(x.to(DEV))


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16'])
def test_dropout_and_droppath_training_on_gpu(precision):
    """SURVEY 8(a15) with the HIP kernels: training with all three rates > 0 against the reference run with forced
    counter-based masks (tests/golden/tiny_dropout.npz); evaluation ignores the rates."""
    z, cfg = load_golden('tiny_trained')
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tiny_dropout.npz'))
    r = [float(v) for v in d['rates']]
    model = build_model(dict(cfg, drop_rate=r[0], attn_drop_rate=r[1], drop_path_rate=r[2]))
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}, strict=True)
    model = model.to(DEV).train()
    model.precision = precision
    model._drop_seed = int(d['base_seed'])
    from motionbert_amd.engine import Engine
    assert not hasattr(Engine, '_torch_attention')      # round 3: the probability dropout runs inside the attention kernels
    x = torch.from_numpy(z['x']).to(DEV).requires_grad_(True)
    out = model(x)
    e_out = rel_l2(out.detach().cpu().numpy(), d['out'])
    (out * torch.from_numpy(z['cot']).to(DEV)).sum().backward()
    e_all, e_worst, worst = grad_errors({n: p.grad.cpu().numpy() for n, p in model.named_parameters()},
                                        {n: d['g.' + n] for n, _ in model.named_parameters()})
    REPORT[f'tiny_dropout.{precision}'] = dict(out=e_out, grad_global=e_all, worst_grad=e_worst, worst_name=worst)
    if precision in ('fp32', 'bf16x3'):
        assert e_out < TOL_FP32 and e_all < TOL_FP32 and e_worst < TOL_FP32, (e_out, e_all, worst, e_worst)
        assert rel_l2(x.grad.cpu().numpy(), d['dx']) < TOL_FP32
    else:
        assert e_out < TOL_BF16_OUT and e_all < TOL_BF16_GRAD, (e_out, e_all)
    del model._drop_seed                       # seeds from torch's generator: two training forwards differ, eval is exact
    with torch.no_grad():
        a, b = model(x), model(x)
        assert not torch.equal(a, b)
        model.eval()
        assert rel_l2(model(x).cpu().numpy(), z['out']) < (TOL_FP32 if precision != 'bf16' else TOL_BF16_OUT)


def test_no_grad_weight_cache_follows_the_parameters():
    """The no-grad path keeps its prepared weights (folded / packed copies) between calls (engine.prepare_weights): a
    second call launches no fold / pack kernels, an in-place parameter update is picked up, and a NEW model that the allocator
    places at the old one's addresses (same version counters) does not hit the old entry."""
    from motionbert_amd import engine, hip_ops
    calls = []
    orig = engine.Engine._prepare_weights

    def counting(self, need_grad):
        calls.append(need_grad)
        return orig(self, need_grad)
    engine.Engine._prepare_weights = counting
    try:
        hip_ops.get().weight_cache.clear()
        x = make_input(2, 27, 17, 21).to(DEV)
        model = build_model(FULL, seed=3)
        trained_like(model, 3)
        model = model.to(DEV).eval()
        with torch.no_grad():
            y0 = model(x)
            y1 = model(x)
            assert len(calls) == 1 and torch.equal(y0, y1)
            model.blocks_st[2].mlp_t.fc1.weight.mul_(1.25)          # in place: the version counter moves
            y2 = model(x)
            assert len(calls) == 2 and not torch.equal(y2, y0)
            sd = {k: v.clone() for k, v in model.state_dict().items()}
            del model
            torch.cuda.synchronize()
            other = build_model(FULL, seed=4)
            trained_like(other, 4)
            other = other.to(DEV).eval()      # very likely the same addresses and versions
            y3 = other(x)
            assert len(calls) == 3
            other.load_state_dict(sd)
            assert torch.equal(other(x), y2) and len(calls) == 4
        # a training step never uses the cache and DROPS it: an update that no version counter sees -- `p.data.mul_()`, what
        # legacy optimizers and EMA code do -- is picked up by the next no-grad call of an ordinary train-then-eval loop
        other.train()
        other(x).sum().backward()
        assert calls[-1] is True and len(calls) == 5
        other.blocks_st[0].mlp_s.fc2.weight.data.mul_(1.5)
        other.eval()
        with torch.no_grad():
            y4 = other(x)
            assert len(calls) == 6 and not torch.equal(y4, y2)
            assert torch.equal(other(x), y4) and len(calls) == 6      # ... and kept again from then on
    finally:
        engine.Engine._prepare_weights = orig
