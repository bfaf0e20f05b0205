"""An input pipeline that can feed the GPU (reference `lib/data/dataset_motion_3d.py:33-67`,
`train.py:219-240`, on-disk format of `tools/convert_h36m.py`: one `%08d.pkl` = {"data_input", "data_label"} per clip).

The reference unpickles one ~99 KB file per clip per access from 12 DataLoader workers, augments per clip in numpy and
collates on the host.  At 8 x 1,000 clips/s that is ~0.8 GB/s of pickle parsing.  Here:

  pack_motion3d()      one-off: the per-clip pickles of a (data_root, subsets, split) become two dense arrays
                       `<prefix>.input.npy` / `<prefix>.label.npy` [N, T, 17, 3] float32 (plain .npy: np.load(mmap_mode='r')).
  PackedMotion3D       memory-maps them.  A batch is one `np.take` of B rows (contiguous 49.6 KB each) straight into a slot
                       of a pinned ring buffer, filled by ONE background thread, copied to the device on a side stream while the
                       previous batch trains; nothing is parsed, nothing is collated.
  on the device        everything `MotionDataset3D.__getitem__` did per clip on the host: random flip of input and label
                       (`flip_data`, utils_data.py:54-66), the synthetic / gt_2d input (`x, y` of the label + confidence 1,
                       dataset_motion_3d.py:49-53), `crop_scale_3d` (utils_data.py:31-52) -- batched torch ops.
  sharding             `rank` / `world`: every rank walks the same per-epoch permutation and takes a strided, equally sized
                       share (what DistributedSampler does), so the N ranks of the data-parallel run never exchange data.
"""
from __future__ import annotations

import json
import os
import pickle
import queue
import threading
from typing import Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from .augment import FLIP_PERM


def pack_motion3d(data_root: str, subset_list: Sequence[str], data_split: str, out_prefix: str) -> dict:
    """Convert the reference's per-clip pickle tree `<data_root>/<subset>/<split>/*.pkl` (file order = the reference's
    `sorted(os.listdir(...))`, dataset_motion_3d.py:19-25) into two dense .npy arrays.  Returns the metadata dict."""
    files = []
    for subset in subset_list:
        d = os.path.join(data_root, subset, data_split)
        files += [os.path.join(d, f) for f in sorted(os.listdir(d))]
    if not files:
        raise ValueError(f'no clips under {data_root} {list(subset_list)} {data_split}')
    with open(files[0], 'rb') as f:
        first = pickle.load(f)
    shape = tuple(np.asarray(first['data_label']).shape)
    has_input = first['data_input'] is not None
    lab = np.lib.format.open_memmap(out_prefix + '.label.npy', mode='w+', dtype=np.float32, shape=(len(files),) + shape)
    inp = np.lib.format.open_memmap(out_prefix + '.input.npy', mode='w+', dtype=np.float32, shape=(len(files),) + shape) if has_input else None
    for i, path in enumerate(files):
        with open(path, 'rb') as f:
            m = pickle.load(f)
        lab[i] = np.asarray(m['data_label'], dtype=np.float32)
        if has_input:
            if m['data_input'] is None:
                raise ValueError(f'{path}: data_input missing (the first clip had one)')
            inp[i] = np.asarray(m['data_input'], dtype=np.float32)
    lab.flush()
    if inp is not None:
        inp.flush()
    meta = dict(n=len(files), clip_shape=list(shape), has_input=bool(has_input), split=data_split, subsets=list(subset_list))
    with open(out_prefix + '.json', 'w') as f:
        json.dump(meta, f)
    return meta


def flip_batch(x: torch.Tensor, which: torch.Tensor) -> torch.Tensor:
    """`flip_data` (utils_data.py:54-66) applied to the clips of the batch where `which` [B] is True."""
    perm = torch.as_tensor(FLIP_PERM, device=x.device)
    f = x.index_select(-2, perm).clone()
    f[..., 0] = -f[..., 0]
    return torch.where(which.view(-1, *([1] * (x.dim() - 1))), f, x)


def crop_scale_3d_batch(motion: torch.Tensor, ratio: torch.Tensor) -> torch.Tensor:
    """`crop_scale_3d` (utils_data.py:31-52) for a batch [B,T,17,3] with one random `ratio` [B] per clip."""
    lo = motion[..., :2].amin(dim=(1, 2))            # [B,2]: xmin, ymin
    hi = motion[..., :2].amax(dim=(1, 2))
    scale = (hi - lo).amax(dim=1) / ratio             # max(xmax-xmin, ymax-ymin) / ratio
    ok = scale != 0
    s = torch.where(ok, scale, torch.ones_like(scale)).view(-1, 1, 1)
    xs = ((lo + hi) - scale.view(-1, 1)) / 2          # [B,2]
    out = motion.clone()
    out[..., :2] = (motion[..., :2] - xs.view(-1, 1, 1, 2)) / s.unsqueeze(-1)
    out[..., 2] = (motion[..., 2] - motion[:, 0:1, 0:1, 2]) / s      # z relative to the first frame's root (:38), same scale
    out = (out - 0.5) * 2
    return torch.where(ok.view(-1, 1, 1, 1), out, torch.zeros_like(out))


def shard_indices(n: int, shuffle: bool, epoch: int, seed: int, rank: int, world: int) -> np.ndarray:
    """The clips of rank `rank` in epoch `epoch`: every rank walks the same permutation and takes a strided share, padded by
    wrap-around to ceil(n / world) clips (what DistributedSampler does).  EQUAL shares are what keeps data-parallel ranks in
    lock-step when the loaders of a pre-training epoch have different lengths (train.py:325-330; `pretrain_epoch_plan`): every
    rank issues the same number of steps per loader, so the gradient all-reduces pair up."""
    order = np.random.default_rng([seed, epoch]).permutation(n) if shuffle else np.arange(n)
    per = -(-n // world)
    return np.resize(order, per * world)[rank::world]


class PackedMotion3D:
    """Memory-mapped packed clips + an asynchronous batch stream.

        ds = PackedMotion3D(prefix, device='cuda', flip=True, synthetic=False, gt_2d=False, scale_range=None)
        for x2d, gt3d in ds.batches(64, shuffle=True, epoch=e, rank=r, world=w):   # device tensors [B,T,17,3]
            ...
    `data_split` semantics follow MotionDataset3D.__getitem__ (dataset_motion_3d.py:42-67): with train=True the input is the
    stored 2D detection with a random flip of input AND label (or, synthetic / gt_2d, the augmented label's x, y + confidence
    1); with train=False the stored input (gt_2d: the label's x, y + confidence 1), no augmentation."""

    def __init__(self, prefix: str, device='cuda', train: bool = True, flip: bool = True, synthetic: bool = False, gt_2d: bool = False,
                 scale_range: Optional[Tuple[float, float]] = None, ring: int = 3):
        with open(prefix + '.json') as f:
            self.meta = json.load(f)
        self.label = np.load(prefix + '.label.npy', mmap_mode='r')
        self.input = np.load(prefix + '.input.npy', mmap_mode='r') if self.meta['has_input'] else None
        self.device = torch.device(device)
        self.train, self.flip, self.synthetic, self.gt_2d, self.scale_range = train, flip, synthetic, gt_2d, scale_range
        self.ring = max(2, int(ring))
        if self.input is None and not (synthetic or gt_2d):
            raise ValueError('Training illegal.')       # dataset_motion_3d.py:59 (no 2D detections and not synthetic / gt_2d)

    def __len__(self):
        return int(self.meta['n'])

    def epoch_indices(self, shuffle: bool, epoch: int, seed: int, rank: int, world: int) -> np.ndarray:
        return shard_indices(len(self), shuffle, epoch, seed, rank, world)

    def _device_stage(self, inp, lab, gen):
        """What MotionDataset3D.__getitem__ did per clip on the host, batched on the device."""
        B = lab.shape[0]
        if not self.train:
            if self.gt_2d:
                inp = torch.cat([lab[..., :2], torch.ones_like(lab[..., :1])], -1)
            return inp, lab
        if self.synthetic or self.gt_2d:
            if self.scale_range is not None:             # Augmenter3D.augment3D (augmentation.py:93-98)
                lo, hi = self.scale_range
                ratio = torch.rand(B, generator=gen, device=lab.device) * (hi - lo) + lo
                lab = crop_scale_3d_batch(lab, ratio)
            if self.flip:
                lab = flip_batch(lab, torch.rand(B, generator=gen, device=lab.device) > 0.5)
            inp = torch.cat([lab[..., :2], torch.ones_like(lab[..., :1])], -1)       # GT x, y and c = 1 (:51-53)
            return inp, lab
        if self.flip:
            which = torch.rand(B, generator=gen, device=lab.device) > 0.5                  # :56-58: input and label together
            inp, lab = flip_batch(inp, which), flip_batch(lab, which)
        return inp, lab

    def batches(self, batch_size: int, shuffle: bool = True, epoch: int = 0, seed: int = 0, rank: int = 0, world: int = 1,
                drop_last: bool = False) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        idx = self.epoch_indices(shuffle, epoch, seed, rank, world)
        chunks = [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]
        if drop_last and chunks and len(chunks[-1]) < batch_size:
            chunks.pop()
        if not chunks:
            return
        cuda = self.device.type == 'cuda'
        shape = (batch_size,) + tuple(self.meta['clip_shape'])
        need_inp = self.input is not None and not (self.synthetic or self.gt_2d)     # otherwise the input is derived from the label
        slots = [tuple(torch.empty(shape, dtype=torch.float32, pin_memory=cuda) for _ in range(2 if need_inp else 1)) for _ in range(self.ring)]
        free, ready = queue.Queue(), queue.Queue(maxsize=self.ring)
        for s in range(self.ring):
            free.put(s)

        stop = threading.Event()

        def producer():                                   # ONE host thread: row gather from the page cache into pinned memory
            try:
                for ch in chunks:
                    s = free.get()
                    if s is None or stop.is_set():        # the consumer went away (break / exception): see the finally below
                        return
                    srt = np.sort(ch)                     # ascending file offsets; the batch is a set, order inside it is irrelevant
                    np.take(self.label, srt, axis=0, out=slots[s][0].numpy()[:len(ch)])
                    if need_inp:
                        np.take(self.input, srt, axis=0, out=slots[s][1].numpy()[:len(ch)])
                    ready.put((s, len(ch)))
                ready.put(None)
            except Exception as e:                        # surface loader errors in the consumer
                ready.put(e)
        th = threading.Thread(target=producer, daemon=True)
        th.start()
        gen = torch.Generator(device=self.device)
        gen.manual_seed((seed * 1000003 + epoch) * 8191 + rank)
        copy_stream = torch.cuda.Stream(self.device) if cuda else None
        pending = None                                    # (slot, lab_dev, inp_dev, event) of the batch in flight
        try:
            while True:
                item = ready.get()
                if isinstance(item, Exception):
                    raise item
                nxt = None
                if item is not None:
                    s, nb = item
                    if cuda:
                        with torch.cuda.stream(copy_stream):
                            lab = slots[s][0][:nb].to(self.device, non_blocking=True)
                            inp = slots[s][1][:nb].to(self.device, non_blocking=True) if need_inp else None
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                    else:
                        lab = slots[s][0][:nb].clone()
                        inp = slots[s][1][:nb].clone() if need_inp else None
                        ev = None
                    nxt = (s, lab, inp, ev)
                if pending is not None:
                    s0, lab0, inp0, ev0 = pending
                    if ev0 is not None:
                        torch.cuda.current_stream(self.device).wait_event(ev0)      # the copy finished before compute touches it ...
                        ev0.synchronize()                                             # ... and before the host refills the pinned slot
                        lab0.record_stream(torch.cuda.current_stream(self.device))
                        if inp0 is not None:
                            inp0.record_stream(torch.cuda.current_stream(self.device))
                    free.put(s0)
                    pending = nxt
                    yield self._device_stage(inp0, lab0, gen)
                else:
                    pending = nxt
                if item is None:
                    break
        finally:
            # also reached when the consumer abandons the generator (break, exception in the training loop, GeneratorExit):
            # release the loader thread -- it may sit in free.get() or in ready.put() on a full queue -- and the pinned slots
            stop.set()
            free.put(None)
            while th.is_alive():
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
            if pending is not None and pending[3] is not None:
                pending[3].synchronize()                   # an H2D copy still reading a pinned slot must finish before it is freed
