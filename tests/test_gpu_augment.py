"""SURVEY 8(f) row 3 on a real MI355X: the 2D augmentation kernel against the oracle (itself pinned to the real Augmenter2D by
tests/golden/augment2d.npz) and the flip-TTA path against two plain forwards."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, make_input

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LITE = dict(dim_in=3, dim_out=3, dim_feat=256, dim_rep=512, depth=5, num_heads=8, mlp_ratio=4, num_joints=17, maxlen=243)


def _aug(z):
    from motionbert_amd.augment import Augmenter2D
    d = z['d2c']
    return Augmenter2D(noise=dict(mean=torch.from_numpy(z['noise_mean']), std=torch.from_numpy(z['noise_std']), weight=torch.from_numpy(z['noise_weight'])),
                       d2c=dict(a=float(d[0]), b=float(d[1]), m=float(d[2]), s=float(d[3])), mask_ratio=0.05, mask_T_ratio=0.1)


@pytest.mark.parametrize('kind,mask,noise', [('noise', False, True), ('mask', True, False), ('both', True, True)])
def test_augment2d_matches_the_reference_fixture(kind, mask, noise):
    z = np.load('tests/golden/augment2d.npz')
    out = _aug(z).augment2D(torch.from_numpy(z['x']).to(DEV), mask=mask, noise=noise, seed=int(z['seed']))
    ref = z['out.' + kind]
    # masks are exact; noise goes through logf / cosf / sqrtf of the device (1e-6-class differences, clipped confidences)
    assert out.shape == ref.shape and float(np.abs(out.cpu().numpy() - ref).max()) < 2e-5
    assert float((out.cpu() == 0).float().mean()) == pytest.approx(float((torch.from_numpy(ref) == 0).float().mean()), abs=1e-6)


def test_augment2d_statistics_at_the_pretraining_shape():
    """[64,243,17,3] (MB_pretrain.yaml): masked fractions and noise scale as configured; a fresh seed per call."""
    z = np.load('tests/golden/augment2d.npz')
    A = _aug(z)
    x = make_input(64, 243, 17, 5).to(DEV)
    y = A.augment2D(x, mask=True, noise=True)
    s1 = A.last_seed
    y2 = A.augment2D(x, mask=True, noise=True)
    assert A.last_seed != s1 and not torch.equal(y, y2)
    # the frame mask is ONE draw per frame shared by the batch (torch.rand(1, T, 1, 1), utils_data augmenter): 243 Bernoulli(0.1) draws,
    # sigma = 0.019 -> a 5-sigma band; the joint mask is one draw per (clip, frame, joint): 2.4e5 draws among the kept frames
    zero = y.abs().sum(-1) == 0                                          # [64, 243, 17]
    frame_off = zero.all(dim=2)                                          # [64, 243]
    assert torch.equal(frame_off, frame_off[:1].expand_as(frame_off))    # shared over the batch
    assert abs(float(frame_off[0].float().mean()) - 0.1) < 0.096
    kept = ~frame_off[0]
    assert abs(float(zero[:, kept].float().mean()) - 0.05) < 0.005
    n = A.augment2D(x, noise=True)
    d = (n[..., :2] - x[..., :2])
    assert 0.002 < float(d.std()) < 0.02 and float(n[..., 2].min()) >= 0 and float(n[..., 2].max()) <= 1
    assert torch.equal(A.augment2D(x), x)                                # neither flag: untouched


def test_two_channel_input_and_cpu_tensor():
    z = np.load('tests/golden/augment2d.npz')
    A = _aug(z)
    x2 = make_input(2, 30, 17, 6)[..., :2].to(DEV)                       # args.no_conf (train.py:163-164)
    assert A.augment2D(x2, mask=True).shape == (2, 30, 17, 2)
    assert A.augment2D(x2, noise=True).shape == (2, 30, 17, 3)           # add_noise synthesises the confidence channel
    with pytest.raises(RuntimeError, match='ROCm device'):
        A.augment2D(x2.cpu(), mask=True)


def test_flip_tta_equals_two_forwards():
    from motionbert_amd.augment import flip_tta
    from oracle.augment_oracle import flip_data
    z = np.load('tests/golden/augment2d.npz')
    assert np.array_equal(flip_data(torch.from_numpy(z['x'])).numpy(), z['flipped'])      # the helper below == reference flip_data
    model = build_model(LITE, seed=0).to(DEV).eval()
    x = make_input(3, 81, 17, 7).to(DEV)
    with torch.no_grad():
        a = model(x)
        b = flip_data(model(flip_data(x)))
        ref = (a + b) / 2
    got = flip_tta(model, x)
    assert got.shape == ref.shape and torch.allclose(got, ref, rtol=0, atol=1e-6)
    got[:, :, 0, :] = 0                                                   # callers write into it (train.py:76)
    with pytest.raises(RuntimeError):
        model.forward(x, return_rep='flip_tta')                                     # not under no_grad
