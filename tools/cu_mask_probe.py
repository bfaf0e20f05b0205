"""Experiment: space-slicing instead of time-slicing for the two block streams.  The st and ts blocks of a level run on two HIP
streams (engine.py), but every kernel fills the whole chip, so the streams mostly take turns.  Here each stream is confined to a
disjoint set of CUs (hipExtStreamCreateWithCUMask), so that a power-bound GEMM of one block really runs beside an HBM-bound
LayerNorm / attention kernel of the other.  Prints ms per train step for several splits.

    python tools/cu_mask_probe.py [--steps 6]
"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from functools import partial


def masked_stream(hip, bits):
    """bits: iterable of CU indices (0..255) this stream may use."""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    if rc != 0:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask failed: {rc}')
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6)
    args = ap.parse_args()
    from motionbert_amd import DSTformer
    from motionbert_amd.engine import Engine
    from motionbert_amd.train import FlatAdamW, pose_loss
    import bench
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **bench.FULL).to(dev)
    model.precision = 'bf16'
    opt = FlatAdamW(model, lr=2e-4, weight_decay=0.01)
    x, gt = bench.make_batch(64, 243, 17, 0, dev)
    hip = ctypes.CDLL('libamdhip64.so')

    def step():
        opt.zero_grad(set_to_none=True)
        total, _ = pose_loss(model(x), gt, bench.LAMBDA_SCALE, bench.LAMBDA_VELOCITY)
        total.backward()
        opt.step()

    def timed(n):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print(f'default streams (no masks): {timed(args.steps):.2f} ms/step', flush=True)
    splits = {
        'halves 0-127 | 128-255': (range(0, 128), range(128, 256)),
        'interleaved even | odd': (range(0, 256, 2), range(1, 256, 2)),
        'main 160 | side 96': (range(0, 160), range(160, 256)),
        'main all | side 128-255': (range(0, 256), range(128, 256)),
    }
    for name, (a, b) in splits.items():
        try:
            main_s, side_s = masked_stream(hip, a), masked_stream(hip, b)
            Engine._side_streams[0] = side_s
            with torch.cuda.stream(main_s):
                ms = timed(args.steps)
            print(f'{name:28s}: {ms:.2f} ms/step', flush=True)
        except Exception as e:
            print(f'{name:28s}: failed: {type(e).__name__}: {e}', flush=True)
    Engine._side_streams.pop(0, None)


if __name__ == '__main__':
    main()
