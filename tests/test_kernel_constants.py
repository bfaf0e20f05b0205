"""The polynomial constants of the packed GELU / GELU' epilogues (motionbert_amd/csrc/gelu_fast.h: gelu_fast2, gelu_fast_grad2),
read from the kernel source and evaluated here in float32 exactly as the kernel does: a typo in one literal would still pass the
bf16-tolerance kernel tests for most inputs, this pins the fp32 error of the formulas themselves (nn.GELU = erf form,
reference lib/model/DSTformer.py:79-85)."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, 'motionbert_amd', 'csrc', 'gelu_fast.h')).read()
F32 = np.float32


def _body(name):
    i = SRC.index(name + '(mbx_f32x2_t u) {')
    return SRC[i:SRC.index('\n}\n', i)]


def _literals(body):
    return [F32(x) for x in re.findall(r'(-?\d+\.\d+(?:e[+-]?\d+)?)f', body)]


def _grid():
    g = np.random.default_rng(0)
    return np.concatenate([np.linspace(-14, 14, 400001), g.normal(0, 1.5, 200000), g.normal(0, 0.05, 50000)]).astype(F32)


def test_gelu_forward_formula():
    """A&S 7.1.28 with 2^(-k/2) folded in: gelu(u) = ((u + a) - a r^16) / 2, a = |u|, r = 1 / (1 + b1 a + ... + b6 a^6)."""
    c = _literals(_body('gelu_fast2'))
    b6, b5, b4, b3, b2, b1, one, half = c
    assert one == F32(1) and half == F32(0.5)
    a_s = [0.0705230784, 0.0422820123, 0.0092705272, 0.0001520143, 0.0002765672, 0.0000430638]      # Abramowitz-Stegun 7.1.28
    for k, b in enumerate([b1, b2, b3, b4, b5, b6]):
        assert abs(float(b) - a_s[k] / 2 ** ((k + 1) / 2)) < 2e-7 * max(1.0, abs(a_s[k])), (k, b)
    u = _grid()
    a = np.abs(u)
    d = (a * b6 + b5).astype(F32)
    for b in (b4, b3, b2, b1, one):
        d = (d * a + b).astype(F32)
    r = (F32(1) / d).astype(F32)
    for _ in range(4):
        r = (r * r).astype(F32)
    g = (((u + a) - a * r) * half).astype(F32)
    exact = 0.5 * u.astype(np.float64) * (1 + erf(u.astype(np.float64) / np.sqrt(2)))
    assert np.abs(g - exact).max() < 1.5e-6


def test_gelu_backward_formula():
    """GELU'(u) = Phi(u) + u phi(u) with erf from A&S 7.1.26 and the Gaussian as exp2(-u^2 / 2 log2 e)."""
    c = _literals(_body('gelu_fast_grad2'))
    p, one, c2, a5, a4, a3, a2, a1, one_b, inv_sqrt_2pi, half, half_b = c
    assert one == one_b == F32(1) and half == half_b == F32(0.5)
    assert abs(float(p) - 0.3275911 / np.sqrt(2)) < 1e-7 and abs(float(c2) + 0.5 * np.log2(np.e)) < 1e-7
    assert abs(float(inv_sqrt_2pi) - 1 / np.sqrt(2 * np.pi)) < 1e-7
    assert [float(x) for x in (a1, a2, a3, a4, a5)] == [float(F32(x)) for x in (0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429)]
    u = _grid()
    a = np.abs(u)
    t = (F32(1) / (a * p + one)).astype(F32)
    gauss = np.exp2(((u * u) * c2).astype(F32)).astype(F32)
    poly = (t * a5 + a4).astype(F32)
    for q in (a3, a2, a1):
        poly = (poly * t + q).astype(F32)
    poly = (poly * t).astype(F32)
    e = (one - poly * gauss).astype(F32)
    se = np.copysign(e, u)
    d = ((u * gauss) * inv_sqrt_2pi + (se * half + half)).astype(F32)
    u64 = u.astype(np.float64)
    exact = 0.5 * (1 + erf(u64 / np.sqrt(2))) + u64 * np.exp(-0.5 * u64 * u64) / np.sqrt(2 * np.pi)
    assert np.abs(d - exact).max() < 1.5e-6
