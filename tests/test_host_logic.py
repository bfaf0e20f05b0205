"""Host sequencing (motionbert_amd/engine.py + model.py autograd node) checked on CPU.

The kernels are replaced by the torch stand-in of tests/mock_ops.py (test infrastructure);
what is verified here is everything *around* the kernels: the order of sub-layers in the two
streams, what is saved, how the residual gradient is threaded, the flat parameter-gradient
layout, and the drop-in module API.  The reference numbers come from tests/golden/*.npz
(minted from the real reference model's autograd)."""
import numpy as np
import pytest
import os

import torch

from motionbert_amd import model as M
from tests.helpers import build_model, load_golden, rel_l2, set_switch
from tests.mock_ops import MockOps


def _load(model, z):
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


@pytest.mark.parametrize('rows', [True, False])
@pytest.mark.parametrize('fold', [True, False])
@pytest.mark.parametrize('name', ['tiny_default', 'tiny_trained'])
def test_engine_fp32_matches_reference_gradients(name, fold, rows):
    """Both backward formulations against the reference's autograd gradients: fold=True is the LayerNorm-folded sequencing
    (LayerNorm backward as the dX GEMM's epilogue from the producers' row dots; the product's bf16 path), fold=False the plain one.
    rows (round 5): inside a Block the folded LayerNorm backward is the epilogue of the row-owner GEMM that takes the row means from its
    own accumulators (mbx_rows_lnbwd_t) -- the producers then run without their row dots; False: the row-dot sequencing everywhere."""
    z, cfg = load_golden(name)
    model = build_model(cfg)
    _load(model, z)
    model.precision, model.fold_ln = 'fp32', fold
    ops = MockOps()
    ops.fuse_rows_lnbwd = ops.fuse_rows_resid_ln = rows
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = M.run(ops, model, x)
    assert out.shape == z['out'].shape and out.dtype == torch.float32
    assert rel_l2(out.detach().numpy(), z['out']) < 2e-6
    (out * torch.from_numpy(z['cot'])).sum().backward()
    assert rel_l2(x.grad.numpy(), z['dx']) < 2e-5
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        assert rel_l2(p.grad.numpy(), z['g.' + n]) < 5e-5, n
    # one prep + the expected number of GEMM launches: per level 2 blocks x 4 sub-layers
    depth = cfg['depth']
    assert ops.calls.count('prep_weights') == 1 and ops.calls.count('fold_norm_weights') == int(fold)
    assert ops.calls.count('gemm_tn') == 8 * 2 * depth + 1
    rw = fold and rows
    if rw:      # the row-owner GEMM takes the row means itself in every sub-layer, and the gradient crosses the Block boundaries in the
        # operand type: the fusion backward of the level below / the embedding backward read the two Blocks' gradients as a pair
        assert ops.calls.count('attn_bwd.0.stats') == ops.calls.count('attn_bwd.1.stats') == 0
        assert ops.calls.count('attn_bwd.0') == ops.calls.count('attn_bwd.1') == 2 * depth
        assert ops.calls.count('fuse_bwd_pair') == depth - 1 and ops.calls.count('fuse_bwd') == 1
        assert ops.calls.count('embed_bwd_pair') == 1 and ops.calls.count('embed_bwd') == 0
    else:
        sfx = '.stats' if fold else ''
        assert ops.calls.count('attn_bwd.0' + sfx) == ops.calls.count('attn_bwd.1' + sfx) == 2 * depth
        assert ops.calls.count('fuse_bwd_pair') == 0 and ops.calls.count('fuse_bwd') == depth
        assert ops.calls.count('embed_bwd_pair') == 0 and ops.calls.count('embed_bwd') == 1
    # folded: the only stand-alone LayerNorm backward left is the final `norm`; every Block LayerNorm runs as a GEMM epilogue
    assert ops.calls.count('layernorm_bwd') == (1 if fold else 8 * depth + 1)
    n_lnbwd = ops.calls.count('gemm_nt.lnbwd') + ops.calls.count('gemm_nt.lnbwd.stream') + ops.calls.count('rows_lnbwd_t')
    assert n_lnbwd == ops.calls.count('unfold_norm_grads') == (8 * depth if fold else 0)
    assert ops.calls.count('lnbwd_rowc') == ((0 if rw else 8 * depth) if fold else 0)
    # gradient stream in the operand type: the inner LayerNorm-backward GEMMs of every Block write no fp32 dx (with the row owners: none does)
    assert ops.calls.count('gemm_nt.lnbwd.stream') + ops.calls.count('rows_lnbwd_t') == ((8 * depth if rw else 6 * depth) if fold else 0)
    assert ops.calls.count('rows_lnbwd_t') == (8 * depth if rw else 0)
    # forward: 8 residual GEMMs per level; with the row-owner kernel the six that are followed by a LayerNorm inside their Block bring
    # its output along (three of the four LayerNorm launches of a Block go)
    assert ops.calls.count('rows_resid_ln') == (6 * depth if rw else 0)
    assert ops.calls.count('gemm_nt.2') + ops.calls.count('rows_resid_ln') == 8 * depth
    assert ops.calls.count('rows_n_pack_many') == (2 if rw else 0)      # one launch for the forward set, one for the backward set
    # stand-alone LayerNorm forwards: 8 per level, minus the two per later level that the fusion kernel of the level before provides
    # (folded: the two Blocks of level 0 share one plain normalisation of the embedding's output, too)
    assert ops.calls.count('layernorm_fwd') == 8 * depth - 2 * (depth - 1) - (6 * depth if rw else 0) - int(fold)
    # the MLPs of a Block (4 per level) save gelu'(u) instead of u where the row-owner tail follows: one-multiply backward epilogue
    assert ops.calls.count('gemm_nt.gelu_d') == ops.calls.count('gemm_nt.mul') == (4 * depth if rw else 0)
    assert ops.calls.count('gemm_nt.1') == (0 if rw else 4 * depth)      # EPI_GELU


@pytest.mark.parametrize('off', ['MBX_BLOCK_GRAD_T', 'MBX_ROWS_RESID_LN', 'MBX_GELU_D', 'MBX_ROWS_LNBWD'])
def test_round5_switches_one_at_a_time(off, monkeypatch):
    """Each A/B switch of round 5 off on its own (the others on): same gradients as the reference, and the launch counts of the sequencing
    it falls back to."""
    z, cfg = load_golden('tiny_trained')
    model = build_model(cfg)
    _load(model, z)
    model.precision, model.fold_ln = 'fp32', True
    set_switch(monkeypatch, off, '0')
    ops = MockOps()
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = M.run(ops, model, x)
    (out * torch.from_numpy(z['cot'])).sum().backward()
    assert rel_l2(out.detach().numpy(), z['out']) < 2e-6 and rel_l2(x.grad.numpy(), z['dx']) < 2e-5
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.numpy(), z['g.' + n]) < 5e-5, n
    depth = cfg['depth']
    c = ops.calls.count
    if off == 'MBX_BLOCK_GRAD_T':       # fp32 across Block boundaries: the first sub-layer of every Block back on the row-dot sequencing
        assert c('fuse_bwd_pair') == c('embed_bwd_pair') == 0 and c('rows_lnbwd_t') == 6 * depth and c('lnbwd_rowc') == 2 * depth
        assert c('gemm_nt.gelu_d') == c('rows_resid_ln') * 4 // 6 == 4 * depth
    elif off == 'MBX_ROWS_RESID_LN':
        assert c('rows_resid_ln') == 0 and c('gemm_nt.2') == 8 * depth and c('layernorm_fwd') == 8 * depth - 2 * (depth - 1) - 1
        assert c('rows_lnbwd_t') == 8 * depth and c('gemm_nt.gelu_d') == 4 * depth
    elif off == 'MBX_GELU_D':
        assert c('gemm_nt.gelu_d') == c('gemm_nt.mul') == 0 and c('gemm_nt.1') == c('gemm_nt.4') == 4 * depth
        assert c('rows_lnbwd_t') == 8 * depth and c('rows_resid_ln') == 6 * depth
    else:                               # no row-owner LayerNorm backward: neither the saved derivative nor the bf16 boundary can follow
        assert c('rows_lnbwd_t') == c('gemm_nt.gelu_d') == c('fuse_bwd_pair') == c('embed_bwd_pair') == 0
        assert c('gemm_nt.lnbwd') + c('gemm_nt.lnbwd.stream') == 8 * depth and c('lnbwd_rowc') == 8 * depth and c('rows_resid_ln') == 6 * depth


@pytest.mark.parametrize('name', ['lite_2x81', 'full_1x243'])
def test_bf16_sequencing_passes_the_fixture_gates_on_cpu(name):
    """The product's DEFAULT bf16 sequencing (every round-5 switch on: row-owner LayerNorm backward and residual GEMM + LayerNorm, saved
    GELU derivative, bf16 gradient across Block boundaries) run by the engine over the torch restatement of the kernel set, against the
    gates of the GPU fixture test (tests/test_gpu_model.py::test_baseline_shape_fixture_fwd_bwd): where a kernel rounds is part of the
    restatement, so a sequencing change that moves the numerics shows up here without a GPU."""
    from tests.helpers import trained_like
    from tests.test_gpu_model import _fixture_grad_errors
    z, cfg = load_golden(name)
    names = [str(n) for n in z['names']]
    ac_per = dict(zip(names, (float(a) for a in z['autocast_grad_per'])))
    ac_out, ac_glob = float(z['autocast_out']), float(z['autocast_grad_global'])
    model = build_model(cfg, seed=0)
    if int(z['trained_seed']) >= 0:
        trained_like(model, int(z['trained_seed']))
    model.precision = 'bf16'
    ops = MockOps()
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = M.run(ops, model, x)
    (out * torch.from_numpy(z['cot'])).sum().backward()
    depth = cfg['depth']
    assert ops.calls.count('rows_lnbwd_t') == 8 * depth and ops.calls.count('rows_resid_ln') == 6 * depth      # the sequencing under test
    assert ops.calls.count('gemm_nt.gelu_d') == 4 * depth and ops.calls.count('fuse_bwd_pair') == depth - 1 and ops.calls.count('embed_bwd_pair') == 1
    e_out = rel_l2(out.detach().numpy(), z['out'])
    e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
    assert e_out < min(2 * ac_out, max(4e-2, ac_out)), (e_out, ac_out)
    assert e_all < min(2 * ac_glob, max(0.08, ac_glob)), (e_all, ac_glob)
    e_dx, ac_dx = rel_l2(x.grad.numpy(), z['dx']), float(z['autocast_dx'])      # (round 6) the input gradient, gated like the rest
    assert e_dx < min(2 * ac_dx, max(0.08, ac_dx)), (e_dx, ac_dx)
    bad = {n: round(per[n], 4) for n in names if per[n] > max(3 * ac_per[n], 0.08)}
    assert not bad, bad


def test_engine_representation_path(golden_dir):
    z, cfg = load_golden('tiny_trained')
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'fp32'
    x = torch.from_numpy(z['x']).requires_grad_(True)
    rep = M.run(MockOps(), model, x, return_rep=True)
    assert rel_l2(rep.detach().numpy(), z['rep']) < 2e-6
    (rep * torch.from_numpy(z['cot_rep'])).sum().backward()
    assert rel_l2(x.grad.numpy(), z['dx_rep']) < 2e-5
    assert model.head.weight.grad is None and model.head.bias.grad is None   # reference autograd leaves them None
    for n, p in model.named_parameters():
        if np.linalg.norm(z['grep.' + n]) > 0 and not n.startswith('head.'):
            assert rel_l2(p.grad.numpy(), z['grep.' + n]) < 5e-5, n


def test_engine_bf16_is_bf16_class():
    """bf16 operands + fp32 accumulation/residual: error must sit at the bf16 noise floor the
    reference itself shows under autocast (7e-3..4e-2), not at 1e-3."""
    z, cfg = load_golden('tiny_trained')
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'bf16'
    x = torch.from_numpy(z['x'])
    out = M.run(MockOps(), model, x)
    e = rel_l2(out.detach().numpy(), z['out'])
    assert 1e-4 < e < 4e-2, e
    (out * torch.from_numpy(z['cot'])).sum().backward()
    worst = max(rel_l2(p.grad.numpy(), z['g.' + n]) for n, p in model.named_parameters() if np.linalg.norm(z['g.' + n]) > 1e-3)
    assert worst < 0.1, worst


def test_no_grad_saves_nothing_and_frozen_params_get_none():
    z, cfg = load_golden('tiny_default')
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'fp32'
    x = torch.from_numpy(z['x'])
    ops = MockOps()
    with torch.no_grad():
        out = M.run(ops, model, x)
    assert not out.requires_grad
    # inference must not pay for training: no transposed weight copies, no backward-only tensors
    assert not any(c.startswith(('gemm_tn', 'attn_bwd')) for c in ops.calls)
    assert ops.last_need_t is False
    out[:, :, 0, :] = 0  # callers write into the output in place (train.py:76, infer_wild.py:82)
    # partial_train_layers (learning.py:69-77) freezes by name
    for n, p in model.named_parameters():
        p.requires_grad = 'head' in n
    out = M.run(MockOps(), model, x)
    out.sum().backward()
    assert model.head.weight.grad is not None and model.joints_embed.weight.grad is None


@pytest.mark.parametrize('name', ['tiny_default', 'tiny_trained'])
def test_no_grad_raw_operand_sequencing(name, monkeypatch):
    """The no-grad path of a Block: 3 launches per attention + MLP pair -- qkv as a row-owner GEMM that makes operand and LayerNorm
    statistics from the fp32 rows (rows_gemm.ln), attention, and ONE kernel for proj + residual + LayerNorm + fc1 + GELU + fc2 + residual
    -- no LayerNorm pass, no bf16 copy of the residual stream; against the reference output and against the training-path sequencing."""
    z, cfg = load_golden(name)
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'fp32'
    x = torch.from_numpy(z['x'])
    ops, ops_plain = MockOps(), MockOps()
    ops_plain.fuse_mlp = False
    with torch.no_grad():
        out = M.run(ops, model, x)
        out_plain = M.run(ops_plain, model, x)
        rep = M.run(MockOps(), model, x, return_rep=True)
    assert rel_l2(out.numpy(), z['out']) < 5e-6 and rel_l2(out_plain.numpy(), z['out']) < 2e-6
    assert rel_l2(rep.numpy(), z['rep']) < 5e-6
    depth = cfg['depth']
    # per level: 2 Blocks x 2 x (qkv from the fp32 rows, attention, proj + residual + MLP in one kernel)
    assert ops.calls.count('proj_mlp_fused_fwd') == ops.calls.count('proj_mlp_pack_weights') == 4 * depth
    assert not any(c.startswith('mlp_fused_fwd') or c == 'mlp_pack_weights' for c in ops.calls)
    assert ops.calls.count('rows_gemm.ln') == ops.calls.count('rows_pack_nk') == 4 * depth
    assert ops.calls.count('gemm_nt.2') == 0                                          # no stand-alone proj + residual GEMM
    # the three-kernel form (proj + residual as its own GEMM, then the fused MLP on its fp32 output) behind the A/B switch
    set_switch(monkeypatch, 'MBX_PROJ_MLP', '0')
    ops3 = MockOps()
    with torch.no_grad():
        out3 = M.run(ops3, model, x)
    set_switch(monkeypatch, 'MBX_PROJ_MLP', '1')
    assert rel_l2(out3.numpy(), z['out']) < 5e-6
    assert ops3.calls.count('mlp_fused_fwd.from_x') == ops3.calls.count('mlp_pack_weights') == 4 * depth and ops3.calls.count('gemm_nt.2') == 4 * depth
    assert ops.calls.count('gemm_nt.0') == ops.calls.count('gemm_nt.1') == 0          # no qkv / fc1 / fc2 launches of the training path
    assert ops.calls.count('layernorm_fwd') == 0                                      # (the final LayerNorm is part of the head kernel)
    assert not any(c.startswith('mlp_fused') for c in ops_plain.calls) and ops_plain.calls.count('gemm_nt.1') == 4 * depth
    # with gradients enabled the training sequencing runs, whatever the provider offers
    ops_g = MockOps()
    M.run(ops_g, model, x.clone().requires_grad_(True)).sum().backward()
    assert not any(c.startswith('mlp_fused') or c.startswith('rows_gemm') or c.startswith('rows_pack') or c.startswith('proj_mlp') for c in ops_g.calls)


def test_average_fusion_variant():
    z, cfg = load_golden('tiny_default')
    from oracle import dstformer_oracle as O
    from tests.helpers import oracle_cfg
    model = build_model(dict(cfg, att_fuse=False))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.') and 'ts_attn' not in k}
    model.load_state_dict(sd, strict=True)
    model.precision = 'fp32'
    x = torch.from_numpy(z['x'])
    out = M.run(MockOps(), model, x)
    ocfg = oracle_cfg(cfg); ocfg.att_fuse = False
    ref, cache = O.forward({k: v.numpy() for k, v in sd.items()}, z['x'], ocfg, want_cache=True)
    assert rel_l2(out.detach().numpy(), ref) < 2e-6
    (out * torch.from_numpy(z['cot'])).sum().backward()
    G, _ = O.backward({k: v.numpy() for k, v in sd.items()}, cache, z['cot'], ocfg)
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.numpy(), G[n]) < 5e-5, n


def _dataparallel_style_replica(module):
    """What torch.nn.parallel.replicate builds per device (torch/nn/parallel/replicate.py): every module is
    `_replicate_for_data_parallel()`-ed (so `_parameters` is EMPTY) and the broadcast parameter copies -- non-leaf
    tensors -- are attached as plain attributes.  Restated here so that the code path runs without two GPUs."""
    mods = list(module.modules())
    copies = [m._replicate_for_data_parallel() for m in mods]
    index = {m: i for i, m in enumerate(mods)}
    for m, r in zip(mods, copies):
        for key, child in m._modules.items():
            r._modules[key] = None if child is None else copies[index[child]]
        for key, param in m._parameters.items():
            if param is not None:
                setattr(r, key, param * 1.0)      # non-leaf copy, gradient flows back to the original (Broadcast does this)
    return copies[0]


def test_dataparallel_replica_runs_and_backpropagates():
    """train.py:256-258 / infer_wild.py:32-34 always wrap the backbone in nn.DataParallel; a replica has no
    `_parameters`, so the fused path must find its 260 tensors by name."""
    z, cfg = load_golden('tiny_trained')
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'fp32'
    rep = _dataparallel_style_replica(model)
    assert not list(rep.named_parameters())
    x = torch.from_numpy(z['x'])
    out = M.run(MockOps(), rep, x)
    assert rel_l2(out.detach().numpy(), z['out']) < 2e-6
    (out * torch.from_numpy(z['cot'])).sum().backward()
    for n, p in model.named_parameters():
        assert p.grad is not None and rel_l2(p.grad.numpy(), z['g.' + n]) < 5e-5, n


def test_in_place_edits_between_forward_and_backward_are_caught():
    z, cfg = load_golden('tiny_default')
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'fp32'
    x = torch.from_numpy(z['x'])
    rep = M.run(MockOps(), model, x, return_rep=True)
    rep.mul_(2.0)                      # backward's tanh' reads this buffer
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        rep.sum().backward()
    out = M.run(MockOps(), model, x)
    with torch.no_grad():
        model.norm.weight.add_(1.0)    # e.g. optimizer.step() before backward
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        out.sum().backward()


def test_parameters_on_the_wrong_device_or_dtype_are_rejected():
    _, cfg = load_golden('tiny_default')
    model = build_model(cfg).double()
    with pytest.raises(RuntimeError, match='fp32 parameters'):
        M.run(MockOps(), model, torch.zeros(1, 4, 17, 3))


def test_engine_bf16x3_is_fp32_class():
    """precision 'bf16x3' (split-operand GEMMs): the host sequencing of the hi/lo planes, checked with the torch
    restatement -- fp32-class agreement with the reference's gradients (the 1e-3 gate with two orders of margin)."""
    z, cfg = load_golden('tiny_trained')
    model = build_model(cfg)
    _load(model, z)
    model.precision = 'bf16x3'
    ops = MockOps()
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = M.run(ops, model, x)
    assert rel_l2(out.detach().numpy(), z['out']) < 2e-5
    (out * torch.from_numpy(z['cot'])).sum().backward()
    assert rel_l2(x.grad.numpy(), z['dx']) < 1e-4
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.numpy(), z['g.' + n]) < 2e-4, n


def test_dropout_and_droppath_match_the_reference_with_forced_masks():
    """training with drop_rate / attn_drop_rate / drop_path_rate > 0.  tests/golden/tiny_dropout.npz is the real
    reference run with its Dropout / DropPath modules drawing the engine's counter-based masks (oracle/make_golden.py
    dropout_fixture): every mask site, tensor layout and scaling must agree, forward and backward."""
    z, cfg = load_golden('tiny_trained')
    d = np.load('tests/golden/tiny_dropout.npz')
    r = [float(v) for v in d['rates']]
    model = build_model(dict(cfg, drop_rate=r[0], attn_drop_rate=r[1], drop_path_rate=r[2]))
    _load(model, z)
    model.precision = 'fp32'
    model.train()
    model._drop_seed = int(d['base_seed'])
    x = torch.from_numpy(z['x']).requires_grad_(True)
    out = M.run(MockOps(), model, x)
    assert rel_l2(out.detach().numpy(), d['out']) < 5e-6
    assert rel_l2(out.detach().numpy(), z['out']) > 0.05, 'dropout must actually change the output'
    (out * torch.from_numpy(z['cot'])).sum().backward()
    assert rel_l2(x.grad.numpy(), d['dx']) < 5e-5
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.numpy(), d['g.' + n]) < 1e-4, n
    model.eval()                                   # evaluation: rates are ignored
    with torch.no_grad():
        assert rel_l2(M.run(MockOps(), model, x.detach()).numpy(), z['out']) < 2e-6


@pytest.mark.parametrize('precision,fold', [('fp32', False), ('bf16x3', False), ('bf16', False), ('fp32', True), ('bf16', True)])
def test_recompute_mode_rebuilds_what_it_does_not_save(precision, fold):
    """model.recompute: LayerNorm outputs and MLP post-activations are rebuilt in backward.  fp32-class modes: bit-identical
    gradients; bf16: the rebuilt GELU starts from the bf16-rounded pre-activation (bf16-level difference).  With LayerNorm
    folding the normalised operand (2 bytes) is what backward keeps INSTEAD of the fp32 sub-layer input (4 bytes), so only the
    post-activations are rebuilt."""
    z, cfg = load_golden('tiny_trained')
    grads, calls = [], []
    for rc in (False, True):
        model = build_model(cfg)
        _load(model, z)
        model.precision, model.recompute, model.fold_ln = precision, rc, fold
        ops = MockOps()
        out = M.run(ops, model, torch.from_numpy(z['x']))
        (out * torch.from_numpy(z['cot'])).sum().backward()
        grads.append({n: p.grad.clone() for n, p in model.named_parameters()})
        calls.append(ops.calls)
    depth = cfg['depth']
    # per level: 2 blocks x 4 sub-layers rebuild their LayerNorm output, 2 blocks x 2 MLPs their post-activation
    assert calls[1].count('layernorm_fwd') == calls[0].count('layernorm_fwd') + (0 if fold else 8 * depth) and calls[1].count('gelu_fwd') == 4 * depth
    worst = max(float((grads[0][n] - grads[1][n]).norm() / grads[0][n].norm().clamp_min(1e-20)) for n in grads[0])
    assert worst == 0.0 if precision != 'bf16' else worst < 2e-2, worst


def test_pretrain_epoch_plan_follows_the_reference_curriculum():
    """train.py:325-330: before `pretrain_3d_curriculum` only 3D batches; afterwards PoseTrack (has_gt), InstaVariety (no gt),
    then 3D -- the same list on every rank, so data-parallel ranks step the loaders in lock-step."""
    from motionbert_amd.train import pretrain_epoch_plan
    assert pretrain_epoch_plan(5, 7, 11, epoch=0) == [('3d', True, True, 11)]
    assert pretrain_epoch_plan(5, 7, 11, epoch=30) == [('posetrack', False, True, 5), ('instav', False, False, 7), ('3d', True, True, 11)]
    assert pretrain_epoch_plan(5, 7, 11, epoch=40, train_2d=False) == [('3d', True, True, 11)]
