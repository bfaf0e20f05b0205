"""the packed input pipeline on CPU (device='cpu' exercises the same ring / thread / sharding logic without a
GPU): content identical to reading the reference's per-clip pickles, MotionDataset3D.__getitem__ semantics, equal shards."""
import os
import pickle

import numpy as np
import pytest
import torch

from motionbert_amd.data import PackedMotion3D, crop_scale_3d_batch, flip_batch, pack_motion3d


@pytest.fixture(scope='module')
def packed(tmp_path_factory):
    root = tmp_path_factory.mktemp('motion3d')
    rng = np.random.default_rng(0)
    clips = {}
    for subset, n in (('H36M-A', 7), ('H36M-B', 6)):
        d = root / subset / 'train'
        d.mkdir(parents=True)
        for i in range(n):
            m = dict(data_input=rng.standard_normal((27, 17, 3)).astype(np.float32), data_label=rng.standard_normal((27, 17, 3)).astype(np.float32))
            with open(d / ('%08d.pkl' % i), 'wb') as f:           # tools/convert_h36m.py:save_clips
                pickle.dump(m, f)
            clips[(subset, i)] = m
    prefix = str(root / 'train_packed')
    meta = pack_motion3d(str(root), ['H36M-A', 'H36M-B'], 'train', prefix)
    order = [('H36M-A', i) for i in range(7)] + [('H36M-B', i) for i in range(6)]       # dataset_motion_3d.py:19-25 file order
    return prefix, meta, [clips[k] for k in order]


def test_pack_keeps_every_clip_in_reference_file_order(packed):
    prefix, meta, clips = packed
    assert meta['n'] == 13 and meta['clip_shape'] == [27, 17, 3] and meta['has_input']
    ds = PackedMotion3D(prefix, device='cpu', train=False)
    assert len(ds) == 13
    got = list(ds.batches(4, shuffle=False))
    assert [b[0].shape[0] for b in got] == [4, 4, 4, 1]
    x = torch.cat([b[0] for b in got]).numpy()
    y = torch.cat([b[1] for b in got]).numpy()
    assert np.array_equal(x, np.stack([c['data_input'] for c in clips])) and np.array_equal(y, np.stack([c['data_label'] for c in clips]))


def test_train_split_flips_input_and_label_together(packed):
    from oracle.augment_oracle import flip_data
    prefix, _, clips = packed
    ds = PackedMotion3D(prefix, device='cpu', train=True, flip=True)
    seen = 0
    for x, y in ds.batches(13, shuffle=False, seed=3):
        ref_x = torch.from_numpy(np.stack([c['data_input'] for c in clips]))
        ref_y = torch.from_numpy(np.stack([c['data_label'] for c in clips]))
        for b in range(13):
            plain = torch.equal(x[b], ref_x[b]) and torch.equal(y[b], ref_y[b])
            flipped = torch.equal(x[b], flip_data(ref_x[b])) and torch.equal(y[b], flip_data(ref_y[b]))
            assert plain or flipped
            seen += flipped
    assert 0 < seen < 13
    g = PackedMotion3D(prefix, device='cpu', train=True, flip=False, gt_2d=True)       # GT x, y and c = 1 (dataset_motion_3d.py:49-53)
    x, y = next(iter(g.batches(13, shuffle=False)))
    assert torch.equal(x[..., :2], y[..., :2]) and float(x[..., 2].min()) == 1.0


def test_crop_scale_matches_the_reference_function():
    z = np.load('tests/golden/augment2d.npz')
    clip, ratio = z['cs_clip'], float(z['cs_ratio'])
    out = crop_scale_3d_batch(torch.from_numpy(np.stack([clip, clip * 0])), torch.tensor([ratio, ratio], dtype=torch.float64))
    assert float(np.abs(out[0].numpy() - z['cs_out']).max()) < 1e-12
    assert float(out[1].abs().max()) == 0.0                                          # scale == 0 -> zeros (utils_data.py:45-46)
    from oracle.augment_oracle import flip_data
    x = torch.randn(3, 5, 17, 3)
    assert torch.equal(flip_batch(x, torch.tensor([True, False, True]))[0], flip_data(x[0])) and torch.equal(flip_batch(x, torch.tensor([True, False, True]))[1], x[1])


def test_shards_are_disjoint_equal_and_cover_the_epoch(packed):
    prefix, _, _ = packed
    ds = PackedMotion3D(prefix, device='cpu', train=False)
    parts = [ds.epoch_indices(True, epoch=2, seed=1, rank=r, world=4) for r in range(4)]
    assert all(len(p) == 4 for p in parts)                                           # ceil(13 / 4): equal work on every rank
    assert set(np.concatenate(parts).tolist()) == set(range(13))
    assert not np.array_equal(parts[0], ds.epoch_indices(True, epoch=3, seed=1, rank=0, world=4))
    n = sum(b[0].shape[0] for b in ds.batches(2, shuffle=True, epoch=2, seed=1, rank=1, world=4, drop_last=True))
    assert n == 4


def test_abandoned_epoch_releases_the_loader_thread(packed):
    """a consumer that stops early (break / exception in the training loop) must not leave the producer thread
    blocked on its queues with the pinned ring alive."""
    import threading
    import gc
    prefix, _, _ = packed
    ds = PackedMotion3D(prefix, device='cpu', train=False, ring=2)
    before = threading.active_count()
    for _ in range(3):
        it = ds.batches(2, shuffle=False)
        next(it)
        it.close()                                   # GeneratorExit at the yield, as a `break` + garbage collection does
        for k, _b in enumerate(ds.batches(2, shuffle=False)):
            if k == 1:
                break
        gc.collect()
    with pytest.raises(ZeroDivisionError):
        for _b in ds.batches(2, shuffle=False):
            1 / 0
    gc.collect()
    assert threading.active_count() == before
    assert sum(b[0].shape[0] for b in ds.batches(4, shuffle=False)) == 13       # and the stream still works afterwards


@pytest.mark.gpu
def test_cuda_stream_delivers_the_stored_clips_bit_for_bit(tmp_path):
    """the CUDA branch (copy stream, events, record_stream, pinned-slot reuse) value-checked: 64 known
    clips, train=False, ring of 2 (so every slot is refilled several times per epoch), two epochs with different orders;
    every delivered batch must equal the stored arrays bit for bit, with a compute kernel still reading the previous batch
    when the next copy lands."""
    n, T = 64, 243
    rng = np.random.default_rng(5)
    prefix = str(tmp_path / 'known')
    arr = {}
    for name in ('input', 'label'):
        a = np.lib.format.open_memmap(f'{prefix}.{name}.npy', mode='w+', dtype=np.float32, shape=(n, T, 17, 3))
        a[:] = rng.standard_normal((n, T, 17, 3)).astype(np.float32)
        a[:, 0, 0, 0] = np.arange(n)                 # clip id in the first element
        a.flush()
        arr[name] = np.array(a)
    import json
    json.dump(dict(n=n, clip_shape=[T, 17, 3], has_input=True, split='test', subsets=['synthetic']), open(prefix + '.json', 'w'))
    ds = PackedMotion3D(prefix, device='cuda', train=False, ring=2)
    inp_d, lab_d = torch.from_numpy(arr['input']).cuda(), torch.from_numpy(arr['label']).cuda()
    busy = torch.randn(4096, 4096, device='cuda')
    for epoch in (0, 1):
        seen, kept = [], []
        for x, y in ds.batches(8, shuffle=True, epoch=epoch, seed=3):
            assert x.is_cuda and y.is_cuda and x.shape == (8, T, 17, 3)
            ids = x[:, 0, 0, 0].long()
            busy = busy @ busy * 1e-4                # keep the compute stream busy: the copies must not race ahead of it
            assert torch.equal(x, inp_d[ids]) and torch.equal(y, lab_d[ids])
            kept.append((x, y, ids))                 # batches stay valid after their pinned slot has been reused
            seen += ids.tolist()
        assert sorted(seen) == list(range(n))
        for x, y, ids in kept:
            assert torch.equal(x, inp_d[ids]) and torch.equal(y, lab_d[ids])
    first = [b[0][:, 0, 0, 0].tolist() for b in ds.batches(8, shuffle=True, epoch=0, seed=3)]
    again = [b[0][:, 0, 0, 0].tolist() for b in ds.batches(8, shuffle=True, epoch=0, seed=3)]
    other = [b[0][:, 0, 0, 0].tolist() for b in ds.batches(8, shuffle=True, epoch=1, seed=3)]
    assert first == again and first != other
    # early exit on the CUDA path: the loader thread and the pinned ring are released
    import threading
    before = threading.active_count()
    for k, _b in enumerate(ds.batches(8, shuffle=False)):
        if k == 2:
            break
    import gc
    gc.collect()
    assert threading.active_count() <= before


@pytest.mark.gpu
def test_pipeline_feeds_the_gpu_faster_than_the_backbone_consumes(tmp_path):
    """[N,243,17,3] clips at the benchmark batch (64): the stream must deliver well above the ~1,300 clips/s the forward pass
    (and ~410 clips/s the training step) consumes on one MI355X, from ONE host thread."""
    import time
    n, T = 4096, 243
    rng = np.random.default_rng(0)
    prefix = str(tmp_path / 'big')
    for name in ('input', 'label'):
        a = np.lib.format.open_memmap(f'{prefix}.{name}.npy', mode='w+', dtype=np.float32, shape=(n, T, 17, 3))
        a[:] = rng.standard_normal((n, 1, 17, 3)).astype(np.float32)
        a.flush()
    import json
    json.dump(dict(n=n, clip_shape=[T, 17, 3], has_input=True, split='train', subsets=['synthetic']), open(prefix + '.json', 'w'))
    ds = PackedMotion3D(prefix, device='cuda', train=True, flip=True)
    for _ in ds.batches(64, epoch=0):      # warm the page cache
        pass
    torch.cuda.synchronize()
    t0, clips = time.perf_counter(), 0
    for x, y in ds.batches(64, epoch=1):
        clips += x.shape[0]
        s = x.sum() + y.sum()                # touch the batch on the compute stream
    torch.cuda.synchronize()
    rate = clips / (time.perf_counter() - t0)
    assert x.is_cuda and x.shape == (64, T, 17, 3) and clips == n
    print(f'packed pipeline: {rate:.0f} clips/s from one loader thread')
    # report only: no wall-clock assertion inside the test suite
