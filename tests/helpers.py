"""Shared helpers for the test-suite (fixtures loading, model construction)."""
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith('cfg.')}
    return z, cfg


def build_model(cfg, seed=None):
    from motionbert_amd import DSTformer
    if seed is not None:
        torch.manual_seed(seed)
    return DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **cfg)


def oracle_cfg(cfg):
    from oracle.dstformer_oracle import OracleConfig
    return OracleConfig(eps=1e-6, **cfg)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def make_input(B, T, J, seed):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, T, J, 2, generator=g) * 2 - 1
    conf = torch.rand(B, T, J, 1, generator=g)
    return torch.cat([xy, conf], -1)


def trained_like(model, seed):
    """Same perturbation as oracle/make_golden.py: non-flat softmaxes, data-dependent fusion."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith('ts_attn'):
                p.add_((torch.randn(p.shape, generator=g) * (0.05 if p.ndim == 2 else 0.2)).to(p.device))
            elif p.ndim >= 2 and 'embed' not in n:
                p.mul_(3.0)
            elif 'norm' in n and n.endswith('weight'):
                p.add_((torch.randn(p.shape, generator=g) * 0.2).to(p.device))
            elif n.endswith('bias'):
                p.add_((torch.randn(p.shape, generator=g) * 0.1).to(p.device))


def set_switch(monkeypatch, name: str, value: str):
    """Flip one MBX_* sequencing switch for the rest of the test: the engine reads them once at import (engine.Switches), so the
    environment variable AND the module's snapshot are patched; both are restored at teardown."""
    import os
    from motionbert_amd import engine
    monkeypatch.setenv(name, value)
    monkeypatch.setattr(engine, 'SWITCHES', engine.Switches.from_env(dict(os.environ)))
