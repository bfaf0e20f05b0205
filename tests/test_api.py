"""Drop-in boundary of lib.model.DSTformer.DSTformer -- CPU-only checks."""
import inspect
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.helpers import build_model, load_golden

# constructor of the reference class, DSTformer.py:270-273
REF_SIGNATURE = [('dim_in', 3), ('dim_out', 3), ('dim_feat', 256), ('dim_rep', 512), ('depth', 5), ('num_heads', 8),
                 ('mlp_ratio', 4), ('num_joints', 17), ('maxlen', 243), ('qkv_bias', True), ('qk_scale', None),
                 ('drop_rate', 0.), ('attn_drop_rate', 0.), ('drop_path_rate', 0.), ('norm_layer', nn.LayerNorm),
                 ('att_fuse', True)]


def test_constructor_signature_matches_reference():
    from motionbert_amd import DSTformer
    sig = inspect.signature(DSTformer.__init__)
    got = [(k, v.default) for k, v in sig.parameters.items() if k != 'self']
    assert got == REF_SIGNATURE
    assert list(inspect.signature(DSTformer.forward).parameters) == ['self', 'x', 'return_rep']
    for meth in ('get_representation', 'get_classifier', 'reset_classifier'):
        assert hasattr(DSTformer, meth)


@pytest.mark.parametrize('name', ['lite', 'full'])
def test_seed0_init_reproduces_reference_weights(name):
    """Same seed -> same weights as the reference class (RNG consumption order, README.md:78-79 sizes)."""
    z, cfg = load_golden('seed0_' + name)
    model = build_model(cfg, seed=0)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(n) for n in z['names']]
    assert sum(v.numel() for v in sd.values()) == int(z['nparam']) == {'lite': 16001549, 'full': 42466317}[name]
    got = np.asarray([[v.double().sum().item(), v.double().abs().sum().item()] for v in sd.values()])
    assert np.allclose(got, z['w_stats'], rtol=1e-9, atol=1e-9)


def test_state_dict_roundtrip_with_dataparallel_prefix():
    z, cfg = load_golden('tiny_default')
    model = build_model(cfg)
    sd = {'module.' + k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    wrapped = nn.DataParallel(model)           # infer_wild.py:32-38 loads 'module.'-prefixed keys strictly
    wrapped.load_state_dict(sd, strict=True)
    assert set(wrapped.state_dict().keys()) == set(sd.keys())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    C = cfg['dim_feat']
    assert shapes['temp_embed'] == (1, cfg['maxlen'], 1, C) and shapes['pos_embed'] == (1, 17, C)
    assert shapes['blocks_st.0.attn_s.qkv.weight'] == (3 * C, C)
    assert shapes['ts_attn.0.weight'] == (2, 2 * C)


def test_load_backbone_overlay_resolves_to_hip_model():
    """lib/utils/learning.py:6 does `from lib.model.DSTformer import DSTformer`; with this repo on the
    path that import must yield the MI355X class and accept load_backbone's exact call (learning.py:83-85)."""
    from lib.model.DSTformer import DSTformer
    import motionbert_amd
    assert DSTformer is motionbert_amd.DSTformer
    args = SimpleNamespace(dim_feat=64, dim_rep=64, depth=1, num_heads=2, mlp_ratio=2, maxlen=16, num_joints=17)
    m = DSTformer(dim_in=3, dim_out=3, dim_feat=args.dim_feat, dim_rep=args.dim_rep, depth=args.depth,
                  num_heads=args.num_heads, mlp_ratio=args.mlp_ratio, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                  maxlen=args.maxlen, num_joints=args.num_joints)
    assert m.ln_eps == 1e-6
    assert float(m.ts_attn[0].weight.abs().sum()) == 0 and float(m.ts_attn[0].bias[0]) == 0.5


def test_cpu_call_fails_loudly_no_fallback():
    _, cfg = load_golden('tiny_default')
    model = build_model(cfg)
    with pytest.raises(RuntimeError, match='no CPU path'):
        model(torch.zeros(1, 4, 17, 3))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 4, 16, 3))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 17, 17, 3))  # T > maxlen=16
    with pytest.raises(RuntimeError, match='parameter container'):
        model.blocks_st[0](torch.zeros(1, 17, 64))


def test_dead_reference_modes_are_rejected():
    from motionbert_amd.model import Attention, Block
    with pytest.raises(NotImplementedError):
        Attention(64, st_mode='coupling')
    with pytest.raises(NotImplementedError):
        Block(64, 2, st_mode='stage_para')
