"""SURVEY 8(f) row 1 on a real MI355X: fused pose loss, flat AdamW and the captured training step."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, load_golden, make_input, rel_l2, trained_like

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LITE = dict(dim_in=3, dim_out=3, dim_feat=256, dim_rep=512, depth=5, num_heads=8, mlp_ratio=4, num_joints=17, maxlen=243)


def _ref_losses(pred, gt, ls, lv):
    """lib/model/loss.py:56-62,81-91,133-142 restated (the reference functions are three-liners); pinned against the REAL
    reference by tests/golden/pose_loss.npz, which oracle/make_golden.py minted with the reference's own loss.py."""
    mpjpe = torch.mean(torch.norm(pred - gt, dim=-1))
    scale = torch.mean(torch.sum(gt * pred, dim=3, keepdim=True), dim=2, keepdim=True) / \
        torch.mean(torch.sum(pred ** 2, dim=3, keepdim=True), dim=2, keepdim=True)
    nm = torch.mean(torch.norm(scale * pred - gt, dim=-1))
    vel = torch.mean(torch.norm((pred[:, 1:] - pred[:, :-1]) - (gt[:, 1:] - gt[:, :-1]), dim=-1)) if pred.shape[1] > 1 else pred.sum() * 0
    return mpjpe, nm, vel, mpjpe + ls * nm + lv * vel


@pytest.mark.parametrize('tag', ['a', 't1', 'b'])
def test_pose_loss_matches_reference_fixture(tag):
    from motionbert_amd.train import pose_loss
    z = np.load('tests/golden/pose_loss.npz')
    pred = torch.from_numpy(z[f'{tag}.pred']).to(DEV).requires_grad_(True)
    gt = torch.from_numpy(z[f'{tag}.gt']).to(DEV)
    total, losses = pose_loss(pred, gt, 0.5, 20.0)
    total.backward()
    got = losses.cpu().numpy().astype(np.float64)
    assert np.allclose(got, z[f'{tag}.losses'], rtol=2e-5, atol=1e-6), (got, z[f'{tag}.losses'])
    assert abs(float(total) - z[f'{tag}.losses'][3]) <= 2e-5 * abs(z[f'{tag}.losses'][3])
    assert rel_l2(pred.grad.cpu().numpy(), z[f'{tag}.dpred']) < 2e-5
    # and the restatement used below agrees with the fixture too
    p64 = torch.from_numpy(z[f'{tag}.pred']).double().requires_grad_(True)
    r = _ref_losses(p64, torch.from_numpy(z[f'{tag}.gt']).double(), 0.5, 20.0)
    assert np.allclose([float(v) for v in r], z[f'{tag}.losses'], rtol=1e-5, atol=1e-7)


def test_pose_loss_general_lambdas_and_scaled_cotangent():
    from motionbert_amd.train import pose_loss
    g = torch.Generator().manual_seed(5)
    pred = (torch.randn(4, 30, 17, 3, generator=g) * 0.5).to(DEV).requires_grad_(True)
    gt = (torch.randn(4, 30, 17, 3, generator=g) * 0.3).to(DEV)
    total, losses = pose_loss(pred, gt, 0.25, 3.0)
    (total * 2.5).backward()
    p64 = pred.detach().double().cpu().requires_grad_(True)
    r = _ref_losses(p64, gt.double().cpu(), 0.25, 3.0)
    (r[3] * 2.5).backward()
    assert rel_l2(pred.grad.cpu().numpy(), p64.grad.numpy()) < 2e-5
    assert np.allclose(losses.cpu().numpy(), [float(v) for v in r], rtol=2e-5)


def test_flat_adamw_matches_torch_adamw():
    """Three steps of FlatAdamW == torch.optim.AdamW on an identical copy (same gradients from the same kernels)."""
    from motionbert_amd.train import FlatAdamW, pose_loss
    a = build_model(LITE, seed=0).to(DEV)
    b = build_model(LITE, seed=0).to(DEV)
    for m in (a, b):
        m.precision = 'fp32'      # bf16 would turn the ulp-level difference of the two AdamW implementations after step 1 into
                                  # bf16 rounding flips in step 2 -- chaotic, 1e-4 .. 3e-4 on the weights from build to build
    x = make_input(2, 27, 17, 3).to(DEV)
    gt = (torch.randn(2, 27, 17, 3, generator=torch.Generator().manual_seed(4)) * 0.3).to(DEV)
    oa = FlatAdamW(a, lr=2e-4, weight_decay=0.01)
    ob = torch.optim.AdamW(b.parameters(), lr=2e-4, weight_decay=0.01)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters())), 'flattening must keep the values'
    for it in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            total, _ = pose_loss(m(x), gt)
            total.backward()
            o.step()
        if it == 1:
            oa.lr = 1e-4
            for g in ob.param_groups:
                g['lr'] = 1e-4
    # Entries whose true gradient is exactly zero (the K third of every qkv bias: softmax is shift invariant) carry pure
    # rounding noise, which Adam normalises to +-lr steps -- two correct implementations differ there by O(lr) after the
    # first step.  So: every entry within 2 lr per step, and every weight MATRIX (real gradients) equal in relative L2.
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p - q).abs().max()) <= 2 * 2e-4 * 3, n
        if p.ndim >= 2 and not n.startswith('ts_attn'):
            assert float((p - q).norm() / q.norm()) < 1e-4, (n, float((p - q).norm() / q.norm()))
    sd = a.state_dict()                           # parameters are views of the flat buffer but still a normal state_dict
    assert len(sd) == 260 and sd['temp_embed'].shape == (1, 243, 1, 256)


def test_flat_adamw_uses_the_backward_buffer_without_copy():
    from motionbert_amd.train import FlatAdamW
    m = build_model(LITE, seed=0).to(DEV)
    opt = FlatAdamW(m, lr=1e-3)
    m(make_input(1, 9, 17, 1).to(DEV)).sum().backward()
    g = opt._flat_grad()
    assert opt._gpack is None and g.data_ptr() == m.head.weight.grad.untyped_storage().data_ptr()
    # a frozen parameter has no gradient: the rest is still the backward's own buffer (no pack), and the range list skips it
    m.head.weight.requires_grad_(False)
    opt.zero_grad(set_to_none=True)
    m(make_input(1, 9, 17, 1).to(DEV)).sum().backward()
    g2 = opt._flat_grad()
    o, k = opt._offs['head.weight']
    assert opt._gpack is None and m.head.weight.grad is None
    assert all(not (lo < o + k and o < hi) for lo, hi in opt._active_ranges()) and len(opt._active_ranges()) == 2


def test_flat_adamw_skips_parameters_without_gradient():
    """torch.optim.AdamW skips a parameter whose .grad is None -- no weight decay, no moment update.
    Frozen layers (partial_train, learning.py:69-77; the reference's optimizer holds only requires_grad parameters,
    train.py:284-289) and the backbone's unused head under get_representation() must stay bit-identical, and the trained
    parameters must move exactly as under torch.optim.AdamW over the trainable subset."""
    from motionbert_amd.train import FlatAdamW
    a = build_model(LITE, seed=2).to(DEV)
    b = build_model(LITE, seed=2).to(DEV)
    for m in (a, b):
        m.precision = 'fp32'
        for n, p in m.named_parameters():      # freeze everything outside the last level, like partial_train_layers
            p.requires_grad_(n.startswith(('blocks_st.4', 'blocks_ts.4', 'ts_attn.4', 'norm.', 'pre_logits', 'head')))
    oa = FlatAdamW(a, lr=1e-3, weight_decay=0.1)
    ob = torch.optim.AdamW([p for p in b.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.1)
    before = {n: p.detach().clone() for n, p in a.named_parameters()}
    x = make_input(2, 9, 17, 5).to(DEV)
    for _ in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            m(x).square().sum().backward()
            o.step()
    moved = 0
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        o, k = oa._offs[n]
        if not p.requires_grad:
            assert torch.equal(p, before[n]), f'frozen parameter {n} changed (weight decay applied without a gradient)'
            assert float(oa.exp_avg[o:o + k].abs().max()) == 0.0 and float(oa.exp_avg_sq[o:o + k].abs().max()) == 0.0, n
        else:
            moved += int(not torch.equal(p, before[n]))
            assert float((p - q).abs().max()) <= 2 * 1e-3 * 3, n
            if p.ndim >= 2 and not n.startswith('ts_attn'):
                assert float((p - q).norm() / q.norm()) < 1e-4, (n, float((p - q).norm() / q.norm()))
    assert moved > 20
    # representation path: head.* gets no gradient (model.py hands back None, like the reference's autograd) -> untouched
    c = build_model(LITE, seed=3).to(DEV)
    oc = FlatAdamW(c, lr=1e-3, weight_decay=0.1)
    hw, hb = c.head.weight.detach().clone(), c.head.bias.detach().clone()
    c.get_representation(x).square().sum().backward()
    oc.step()
    assert torch.equal(c.head.weight, hw) and torch.equal(c.head.bias, hb) and float(oc.state_t[0]) == 1.0
    # nothing has a gradient -> no step at all
    oc.zero_grad(set_to_none=True)
    oc.step()
    assert float(oc.state_t[0]) == 1.0


def test_flat_adamw_state_dict_is_torch_format():
    """optimizer.state_dict() in the reference's checkpoints (train.py save_checkpoint / resume) is torch.optim.AdamW's
    {'state', 'param_groups'}: a torch AdamW state loads into FlatAdamW and continues identically, FlatAdamW's own state
    round-trips (copies, not live tensors) and a state for another model is refused."""
    from motionbert_amd.train import FlatAdamW
    a = build_model(LITE, seed=4).to(DEV)
    b = build_model(LITE, seed=4).to(DEV)
    for m in (a, b):
        m.precision = 'fp32'
    x = make_input(2, 9, 17, 6).to(DEV)
    ob = torch.optim.AdamW(b.parameters(), lr=5e-4, weight_decay=0.01)
    for _ in range(2):
        ob.zero_grad(set_to_none=True)
        b(x).square().sum().backward()
        ob.step()
    with torch.no_grad():
        for p, q in zip(a.parameters(), b.parameters()):
            p.copy_(q)
    oa = FlatAdamW(a, lr=1e-3, weight_decay=0.5)
    oa.load_state_dict(ob.state_dict())                     # torch format in
    assert float(oa.state_t[0]) == 2.0 and oa.lr == 5e-4 and oa.param_groups[0]['weight_decay'] == 0.01
    for m, o in ((a, oa), (b, ob)):
        o.zero_grad(set_to_none=True)
        m(x).square().sum().backward()
        o.step()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p - q).abs().max()) <= 2 * 5e-4, n
        if p.ndim >= 2 and not n.startswith('ts_attn'):
            assert float((p - q).norm() / q.norm()) < 1e-5, n
    sd = oa.state_dict()                                     # torch format out: loads into torch.optim.AdamW
    assert set(sd) >= {'state', 'param_groups'} and len(sd['state']) == 260 and sd['param_groups'][0]['params'] == list(range(260))
    ob2 = torch.optim.AdamW(b.parameters(), lr=1.0)
    ob2.load_state_dict({k: v for k, v in sd.items() if k != 'names'})
    assert ob2.param_groups[0]['lr'] == 5e-4 and float(ob2.state[next(iter(b.parameters()))]['step']) == 3.0
    keep = sd['state'][5]['exp_avg'].clone()
    oa.exp_avg.add_(1.0)
    assert torch.equal(sd['state'][5]['exp_avg'], keep), 'state_dict() must return copies'
    oa.load_state_dict(sd)
    o5, k5 = oa._offs[oa._names[5]]
    assert torch.equal(oa.exp_avg[o5:o5 + k5].view(keep.shape), keep)
    other = FlatAdamW(build_model(dict(LITE, depth=2), seed=0).to(DEV), lr=1e-3)
    with pytest.raises(ValueError):
        other.load_state_dict(sd)
    with pytest.raises(KeyError):
        oa.load_state_dict(dict(exp_avg=oa.exp_avg))


def test_graphed_train_step_refuses_dropout():
    from motionbert_amd.train import FlatAdamW, GraphedTrainStep
    m = build_model(dict(LITE, depth=1, drop_path_rate=0.1), seed=0).to(DEV)
    with pytest.raises(NotImplementedError, match='dropout'):
        GraphedTrainStep(m, FlatAdamW(m), make_input(1, 9, 17, 1).to(DEV), torch.zeros(1, 9, 17, 3, device=DEV))


def test_graphed_train_step_matches_eager_steps():
    """forward + loss + backward + AdamW replayed from one hipGraph == the same steps issued eagerly."""
    from motionbert_amd.train import FlatAdamW, GraphedTrainStep, pose_loss
    a = build_model(LITE, seed=1).to(DEV)
    b = build_model(LITE, seed=1).to(DEV)
    oa, ob = FlatAdamW(a, lr=2e-4, weight_decay=0.01), FlatAdamW(b, lr=2e-4, weight_decay=0.01)
    batches = [(make_input(2, 27, 17, 10 + i).to(DEV), (torch.randn(2, 27, 17, 3, generator=torch.Generator().manual_seed(20 + i)) * 0.3).to(DEV))
               for i in range(3)]
    step = GraphedTrainStep(a, oa, *batches[0])
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters())), 'capture must not change the training state'
    for i, (x, gt) in enumerate(batches):
        la = step(x, gt)
        ob.zero_grad(set_to_none=True)
        total, lb = pose_loss(b(x), gt)
        total.backward()
        ob.step()
        assert torch.allclose(la, lb, rtol=1e-6, atol=0), (i, la, lb)
        if i == 0:
            oa.lr = 1e-4                      # device-side learning rate: no re-capture
            ob.lr = 1e-4
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters())), 'graph replay and eager steps must be bit-identical'
    assert float(oa.state_t[0]) == 3.0


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5: the pre-training step with its 2D branch, the ActionNet step
# ------------------------------------------------------------------------------------------------------------------
def _ref_loss_2d(pred, target, conf):
    """lib/model/loss.py:72-77 restated; pinned against the REAL reference by tests/golden/loss_2d.npz."""
    return torch.mean(torch.norm((pred[..., :2] - target[..., :2]) * conf, dim=-1))


@pytest.mark.parametrize('tag', ['a', 'b', 'z'])
def test_loss_2d_weighted_matches_reference_fixture(tag):
    from motionbert_amd.train import loss_2d_weighted
    z = np.load('tests/golden/loss_2d.npz')
    batch = torch.from_numpy(z[f'{tag}.batch']).to(DEV)
    pred = torch.from_numpy(z[f'{tag}.pred']).to(DEV).requires_grad_(True)
    target = batch - batch[:, :, 0:1, :]                      # rootrel, train.py:165-166
    loss = loss_2d_weighted(pred, target, batch[..., 2:])     # confidence = a strided view of the batch (no copy)
    (loss * 1.5).backward()
    assert abs(float(loss) - float(z[f'{tag}.loss'])) <= 2e-5 * float(z[f'{tag}.loss'])
    assert rel_l2(pred.grad.cpu().numpy(), 1.5 * z[f'{tag}.dpred']) < 2e-5
    assert float(pred.grad[..., 2].abs().max()) == 0.0
    p64 = torch.from_numpy(z[f'{tag}.pred']).double()
    b64 = torch.from_numpy(z[f'{tag}.batch']).double()
    assert abs(float(_ref_loss_2d(p64, b64 - b64[:, :, 0:1], b64[..., 2:])) - float(z[f'{tag}.loss'])) < 1e-6
    with pytest.raises(RuntimeError):
        loss_2d_weighted(pred, target[:, :, :5], batch[..., 2:])


def _aug_from_fixture():
    from motionbert_amd.augment import Augmenter2D
    z = np.load('tests/golden/augment2d.npz')
    d = z['d2c']
    return Augmenter2D(noise=dict(mean=torch.from_numpy(z['noise_mean']), std=torch.from_numpy(z['noise_std']), weight=torch.from_numpy(z['noise_weight'])),
                       d2c=dict(a=float(d[0]), b=float(d[1]), m=float(d[2]), s=float(d[3])), mask_ratio=0.05, mask_T_ratio=0.1)


def test_pretrain_step_matches_the_restated_reference_loop():
    """train.py:155-206 (MB_pretrain.yaml: rootrel, mask + noise, lambda_scale 0.5, lambda_3d_velocity 20): one PoseTrack-like 2D
    batch (T=30, has_gt), one InstaVariety-like 2D batch (T=81, no noise), one 3D batch, through PretrainStep on model `a`
    and through the reference's statements restated with torch ops (same augmented input via the same seed, the reference's
    loss formulas, FlatAdamW) on model `b`: same losses, same parameters afterwards."""
    from motionbert_amd.train import FlatAdamW, PretrainStep
    a = build_model(LITE, seed=6).to(DEV)
    b = build_model(LITE, seed=6).to(DEV)
    for m in (a, b):
        m.precision = 'fp32'
    oa, ob = FlatAdamW(a, lr=5e-4, weight_decay=0.01), FlatAdamW(b, lr=5e-4, weight_decay=0.01)
    aug = _aug_from_fixture()
    step = PretrainStep(a, oa, aug=aug, rootrel=True, mask=True, noise=True, lambda_scale=0.5, lambda_velocity=20.0)
    g = torch.Generator().manual_seed(9)
    batches = [(make_input(3, 30, 17, 71).to(DEV), None, False, True), (make_input(2, 81, 17, 72).to(DEV), None, False, False),
               (make_input(2, 27, 17, 73).to(DEV), (torch.randn(2, 27, 17, 3, generator=g) * 0.3).to(DEV), True, True)]
    for k, (x, gt, has_3d, has_gt) in enumerate(batches):
        gt = x if gt is None else gt                          # the 2D datasets return (motion_2d, motion_2d)
        x0 = x.clone()
        la = step(x, gt, has_3d=has_3d, has_gt=has_gt, seed=1000 + k)
        assert torch.equal(x, x0), 'the step must not modify the batch it was handed'
        # ---- the reference's statements
        conf = x[..., 2:].clone()
        tgt = gt - gt[:, :, 0:1, :]
        xin = aug.augment2D(x, noise=has_gt, mask=True, seed=1000 + k)
        pred = b(xin)
        ob.zero_grad(set_to_none=True)
        if has_3d:
            r = _ref_losses(pred, tgt, 0.5, 20.0)
            total, lb = r[3], torch.stack([v.detach() for v in r])
        else:
            total = _ref_loss_2d(pred, tgt, conf)
            lb = torch.stack([total.detach() * 0] * 3 + [total.detach()])
        total.backward()
        ob.step()
        assert torch.allclose(la, lb.float(), rtol=2e-5, atol=1e-7), (k, la, lb)
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p - q).abs().max()) <= 2 * 5e-4 * 3, n
        if p.ndim >= 2 and not n.startswith('ts_attn'):
            assert float((p - q).norm() / q.norm()) < 1e-4, (n, float((p - q).norm() / q.norm()))
    with pytest.raises(NotImplementedError):
        PretrainStep(a, oa, aug=aug, lambda_lv=0.1)


def test_action_step_two_lr_groups_match_torch_adamw():
    """train_action.py:143-149,172-188: AdamW with the backbone at lr_backbone and the head at lr_head (MB_ft_NTU60_xsub.yaml:
    7-9), cross-entropy on ActionNet scores -- ActionStep (two flat optimizers, fused pooling tail, unused backbone head
    untouched) against torch.optim.AdamW with the same two parameter groups on an identical copy."""
    from motionbert_amd.action import ActionNet
    from motionbert_amd.train import ActionStep
    cfg = dict(LITE, depth=2, dim_feat=128, dim_rep=128, num_heads=4)      # head dim 32 (the kernels take 32 or 64)

    def mk():
        torch.manual_seed(91)
        net = ActionNet(backbone=build_model(cfg), dim_rep=128, num_classes=60, dropout_ratio=0., version='class', hidden_dim=2048, num_joints=17).to(DEV)
        net.backbone.precision = 'fp32'
        return net.train()
    a, b = mk(), mk()
    step = ActionStep(a, lr_backbone=1e-4, lr_head=1e-3, weight_decay=0.01)
    ob = torch.optim.AdamW([{'params': [p for p in b.backbone.parameters() if p.requires_grad], 'lr': 1e-4},
                            {'params': list(b.head.parameters()), 'lr': 1e-3}], lr=1e-4, weight_decay=0.01)
    x = torch.stack([make_input(2, 27, 17, 80 + i) for i in range(4)]).to(DEV)           # [N=4, M=2, T, 17, 3]
    labels = torch.tensor([3, 7, 59, 0], device=DEV)
    hw = a.backbone.head.weight.detach().clone()
    for it in range(3):
        la, _ = step(x, labels)
        ob.zero_grad(set_to_none=True)
        lb = torch.nn.functional.cross_entropy(b(x), labels)
        lb.backward()
        ob.step()
        assert abs(float(la) - float(lb)) < 1e-4 * abs(float(lb)), (it, float(la), float(lb))
        if it == 1:
            step.decay(0.99)
            for gk in ob.param_groups:
                gk['lr'] *= 0.99
    assert torch.equal(a.backbone.head.weight, hw), 'the backbone head is unused on the representation path: no decay, no update'
    assert torch.equal(b.backbone.head.weight, hw)
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        lr = 1e-3 if n.startswith('head.') else 1e-4
        assert float((p - q).abs().max()) <= 2 * lr * 3, n
        if p.ndim >= 2 and 'ts_attn' not in n:
            assert float((p - q).norm() / q.norm()) < 1e-4, (n, float((p - q).norm() / q.norm()))
