"""Attention kernels alone at the bench shape (64 clips x 243 frames x 17 joints, 8 heads of 64): HIP-event time per launch,
algorithmic HBM bytes and the rate they imply.  Also the command to put under `rocprofv3 --pmc` for the LDS / MFMA counters.
    python tools/attn_bench.py [--iters 20] [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops
from motionbert_amd.engine import MODE_SPATIAL, MODE_TEMPORAL


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--frames', type=int, default=243)
    a = ap.parse_args()
    ops = hip_ops.get()
    B, T, J, H, hd = a.batch, a.frames, 17, 8, 64
    C, M = H * hd, B * T * J
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn(M, 3 * C, device=dev, generator=g).to(torch.bfloat16)
    d_o = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    o, lse = torch.empty(M, C, device=dev, dtype=torch.bfloat16), torch.empty(M, H, device=dev)
    dqkv = torch.empty_like(qkv)
    scale = hd ** -0.5
    for name, mode in (('spatial', MODE_SPATIAL), ('temporal', MODE_TEMPORAL)):
        fwd = lambda: ops.attn_fwd(qkv, o, lse, B, T, J, H, scale, mode)
        bwd = lambda: ops.attn_bwd(qkv, o, d_o, lse, dqkv, B, T, J, H, scale, mode)
        # backward: qkv, dO, dqkv, lse -- and O (delta = dO . O) in the temporal kernels; the one-wave kernel takes delta from P o dP
        bwd_bytes = M * (3 * C + C + 3 * C + (0 if mode == MODE_SPATIAL else C)) * 2 + M * H * 4
        for tag, fn, nbytes in (('fwd', fwd, M * (3 * C + C) * 2 + M * H * 4), ('bwd', bwd, bwd_bytes)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f'{name:8s} {tag}: {ms:.4f} ms   algorithmic {nbytes / 1e9:.2f} GB -> {nbytes / ms / 1e9:.2f} TB/s', flush=True)


if __name__ == '__main__':
    main()
