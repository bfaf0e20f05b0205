"""The numpy oracle must reproduce the fixtures minted from the real reference model
(oracle/make_golden.py).  Runs on CPU, needs neither a GPU nor /root/reference."""
import numpy as np
import pytest

from oracle import dstformer_oracle as O
from tests.helpers import load_golden, oracle_cfg, rel_l2


@pytest.mark.parametrize('name', ['tiny_default', 'tiny_trained'])
def test_oracle_matches_reference_tiny(name):
    z, cfg = load_golden(name)
    ocfg = oracle_cfg(cfg)
    P = {k[2:]: z[k] for k in z.files if k.startswith('w.')}
    out, cache = O.forward(P, z['x'], ocfg, want_cache=True)
    assert rel_l2(out, z['out']) < 1e-12
    assert rel_l2(cache['rep'].reshape(z['rep'].shape), z['rep']) < 1e-12
    G, dx = O.backward(P, cache, z['cot'], ocfg)
    assert rel_l2(dx, z['dx']) < 1e-10
    for k in P:
        ref = z['g.' + k]
        # fixture gradients are stored as fp32
        assert rel_l2(G[k], ref) < 5e-7, k
    # representation path (ActionNet caller): head receives no gradient
    out_r, cache = O.forward(P, z['x'], ocfg, return_rep=True, want_cache=True)
    G, dx = O.backward(P, cache, z['cot_rep'], ocfg, return_rep=True)
    assert rel_l2(dx, z['dx_rep']) < 1e-10
    assert np.all(G['head.weight'] == 0) and np.all(z['grep.head.weight'] == 0)
    for k in P:
        if np.linalg.norm(z['grep.' + k]) > 0:
            assert rel_l2(G[k], z['grep.' + k]) < 5e-7, k


def test_oracle_elementary_ops_are_consistent():
    """finite-difference check of the hand-written backward formulas (independent of the reference)."""
    rng = np.random.default_rng(0)
    B, T, J, H, hd = 2, 5, 17, 2, 4
    C = H * hd
    qkv = rng.standard_normal((B * T, J, 3 * C))
    do = rng.standard_normal((B * T, J, C))
    for mode in ('spatial', 'temporal'):
        o, p = O.attention_fwd(qkv, B, T, J, H, 0.37, mode)
        g = O.attention_bwd(do, qkv, p, B, T, J, H, 0.37, mode)
        idx = tuple(rng.integers(0, s) for s in qkv.shape)
        e = np.zeros_like(qkv); e[idx] = 1e-6
        fd = ((O.attention_fwd(qkv + e, B, T, J, H, 0.37, mode)[0] - O.attention_fwd(qkv - e, B, T, J, H, 0.37, mode)[0]) * do).sum() / 2e-6
        assert abs(fd - g[idx]) < 1e-6 * max(1, abs(fd))
    u = rng.standard_normal(100)
    fd = (O.gelu_fwd(u + 1e-6) - O.gelu_fwd(u - 1e-6)) / 2e-6
    assert np.allclose(fd, O.gelu_grad(u), atol=1e-8)


@pytest.mark.timeout(600)
def test_oracle_matches_reference_at_baseline_config0():
    """BASELINE.json configs[0] (MotionBERT-Lite, [2,81,17,3]): the oracle's forward AND hand-written backward against the
    reference's fp64 autograd stored in tests/golden/lite_2x81.npz (full gradients of the small tensors, a fixed
    4096-element sample of every large one, the l2 norm of all 260)."""
    import torch
    from tests.helpers import build_model, trained_like
    z, cfg = load_golden('lite_2x81')
    model = build_model(cfg, seed=0)
    if int(z['trained_seed']) >= 0:
        trained_like(model, int(z['trained_seed']))
    P = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    ocfg = oracle_cfg(cfg)
    out, cache = O.forward(P, z['x'], ocfg, want_cache=True)
    assert rel_l2(out, z['out']) < 1e-10
    G, dx = O.backward(P, cache, z['cot'], ocfg)
    assert rel_l2(dx, z['dx']) < 5e-7          # stored as fp32
    for k, n in enumerate(str(n) for n in z['names']):
        g = G[n].reshape(-1)
        ref_l2 = z['g_stats'][k, 0]
        assert abs(np.linalg.norm(g) - ref_l2) <= 1e-9 * max(ref_l2, 1e-12), n
        if 'g.' + n in z.files:
            assert rel_l2(g, z['g.' + n]) < 5e-7 or ref_l2 < 1e-12, n
        else:
            assert rel_l2(g[z[f'idx.{g.size}']], z['gs.' + n]) < 5e-7, n


@pytest.mark.parametrize('name', ['tiny_default', 'tiny_trained'])
def test_plain_torch_restatement_matches_the_fixtures(name):
    """oracle/torch_model.py (the CPU-baseline port of bench.py: the reference's operator mix as one plain torch function) against
    the reference-minted fixtures: output, representation, input gradient and every parameter gradient through autograd."""
    import torch
    from oracle import torch_model as TM
    z, cfg = load_golden(name)
    P = {k[2:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith('w.')}
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = TM.forward(P, x, cfg['depth'], cfg['num_heads'])
    assert rel_l2(out.detach().numpy(), z['out']) < 2e-6
    assert rel_l2(TM.forward(P, x, cfg['depth'], cfg['num_heads'], return_rep=True).detach().numpy(), z['rep']) < 2e-6
    (out * torch.from_numpy(z['cot'])).sum().backward()
    assert rel_l2(x.grad.numpy(), z['dx']) < 2e-5
    for n, p in P.items():
        assert rel_l2(p.grad.numpy(), z['g.' + n]) < 5e-5, n
