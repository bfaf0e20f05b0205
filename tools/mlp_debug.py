"""Localise an error of mbx_mlp_fused_fwd: structured cases (epilogue only / constant hidden / one chunk / general) and the error
per 32-row x 32-column block of one 128-row tile.  `python tools/mlp_debug.py [C]`"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402

BF = torch.bfloat16
torch.set_printoptions(linewidth=220, precision=3, sci_mode=False)


def run(ops, a, w1, w2, b1, b2, x, raw=0):
    M, C = a.shape
    y = torch.full((M, C), float('nan'), device='cuda')
    packed = ops.mlp_pack_weights(w1, w2)
    ops.mlp_fused_fwd(a, raw, packed, b1, b2, w1.float().sum(1), x, y, None, 1e-6, None, None)
    torch.cuda.synchronize()
    u = a.float() @ w1.float().t() + b1
    ref = x + F.gelu(u).to(BF).float() @ w2.float().t() + b2
    return y, ref


def grid(y, ref, M=128):
    d = (y - ref)[:M].abs()
    C = d.shape[1]
    g = d.reshape(M // 32, 32, C // 32, 32).amax((1, 3))
    return g


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ops = hip_ops.get()
    dev = 'cuda'
    gen = torch.Generator(device=dev).manual_seed(0)
    M = 128
    x = torch.randn(M, C, device=dev, generator=gen)
    a = torch.randn(M, C, device=dev, generator=gen).to(BF)
    for hidden in (64, 128, 1024):
        w1 = (torch.randn(hidden, C, device=dev, generator=gen) * 0.05).to(BF)
        w2 = (torch.randn(C, hidden, device=dev, generator=gen) * 0.05).to(BF)
        b1, b2 = torch.randn(hidden, device=dev, generator=gen) * 0.3, torch.randn(C, device=dev, generator=gen) * 0.3
        z1, zb1 = torch.zeros_like(w1), torch.zeros_like(b1)
        for name, (ww1, bb1) in (('epilogue only (w1 = 0, b1 = 0)', (z1, zb1)), ('constant hidden (w1 = 0)', (z1, torch.full_like(b1, 0.7))),
                                 ('per-column constant hidden (w1 = 0, b1 random)', (z1, b1)), ('general', (w1, b1))):
            y, ref = run(ops, a, ww1, w2, bb1, b2, x)
            e = float((y - ref).norm() / (ref - x).norm().clamp_min(1e-6))
            print(f'C={C} hidden={hidden} {name}: rel err of the branch {e:.3e}, nan {int(torch.isnan(y).sum())}')
            if e > 1e-3:
                print(grid(y, ref).cpu())
        # which hidden columns are wrong?  one-hot w2 rows pick single hidden units: y[:, n] = x + g[:, perm[n]] (first C hidden units)
        if hidden >= 64:
            w2e = torch.zeros(C, hidden, device=dev)
            idx = torch.arange(min(C, hidden), device=dev)
            w2e[idx, idx] = 1.0
            y, ref = run(ops, a, w1, w2e.to(BF), b1, torch.zeros_like(b2), x)
            bad = ((y - ref).abs().amax(0) > 2e-2).nonzero().flatten().tolist()
            print(f'C={C} hidden={hidden} identity fc2: wrong output columns (= hidden units) {bad[:64]}{" ..." if len(bad) > 64 else ""} ({len(bad)} of {min(C, hidden)})')
            if bad:
                n = bad[0]
                print('  first wrong column', n, 'kernel', (y - x)[:4, n].tolist(), 'reference', (ref - x)[:4, n].tolist())
                g_ref = (ref - x)[:, :min(C, hidden)]
                got = (y - x)[:, n]
                match = ((g_ref - got[:, None]).abs().amax(0) < 2e-2).nonzero().flatten().tolist()
                print('  the kernel wrote there the reference value of hidden unit(s)', match[:8])


if __name__ == '__main__':
    main()
