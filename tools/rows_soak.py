"""Soak of the N-resident row-owner kernels (csrc/gemm_rows_n.hip): every launch on the same inputs must give bit-identical outputs -- a
counted wait that is one too weak (or a register reused under a load in flight) shows up as a rare mismatch, not as a wrong test value.
    python tools/rows_soak.py [iterations] [clips]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M, dev, BF = clips * 243 * 17, 'cuda', torch.bfloat16
ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
# dim_feat 512 (16 column tiles per wave) and 256 (8 tiles, round 6: MotionBERT-Lite) with the contraction lengths the models have
for N, K in ((512, 512), (512, 1024), (512, 1536), (256, 256), (256, 768), (256, 1024)):
    a = (torch.randn(M, K, device=dev, generator=g) * 0.6).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
    pk = ops.rows_n_pack(w)
    xhat = torch.randn(M, N, device=dev, generator=g).to(BF)
    rstd = torch.rand(M, device=dev, generator=g) + 0.5
    dres = torch.randn(M, N, device=dev, generator=g).to(BF)
    bias = torch.randn(N, device=dev, generator=g)
    resid = torch.randn(M, N, device=dev, generator=g)
    ref_dx = torch.empty(M, N, device=dev, dtype=BF)
    ops.rows_lnbwd_t(a, pk, xhat, rstd, dres, ref_dx)
    ref = [torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, device=dev), torch.empty(M, device=dev)]
    ops.rows_resid_ln(a, pk, bias, resid, *ref, 1e-6)
    # EVERY row against torch, in slabs (round 6: a token fragment that was read before it had landed showed up as a handful of wrong rows
    # among 264,384 -- and identically from launch to launch); the worst ROW decides, not the norm over all of them
    worst = [0.0, 0.0, 0.0]
    for r0 in range(0, M, 32768):
        rows = slice(r0, min(M, r0 + 32768))
        acc = a[rows].float() @ w.float().t()
        xh = xhat[rows].float()
        want_dx = dres[rows].float() + rstd[rows, None] * (acc - acc.mean(-1, keepdim=True) - xh * (acc * xh).mean(-1, keepdim=True))
        yy = resid[rows] + acc + bias
        mu = yy.mean(-1, keepdim=True)
        want_xh = (yy - mu) * torch.rsqrt(((yy - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
        for k, (got, want) in enumerate(((ref_dx[rows].float(), want_dx), (ref[0][rows], yy), (ref[1][rows].float(), want_xh))):
            worst[k] = max(worst[k], float(((got - want).norm(dim=-1) / want.norm(dim=-1)).max()))
    e = worst
    print(f'N = {N}, K = {K:4d}: worst row of {M} against torch: dx {e[0]:.2e}  y {e[1]:.2e}  xhat {e[2]:.2e}', flush=True)
    bad += int(e[0] > 1.2e-2 or e[1] > 1e-5 or e[2] > 1.2e-2)
    dx = torch.empty_like(ref_dx)
    out = [torch.empty_like(t) for t in ref]
    n_bad = [0, 0]
    for it in range(iters):
        dx.fill_(7.0)
        ops.rows_lnbwd_t(a, pk, xhat, rstd, dres, dx)
        n_bad[0] += int(not torch.equal(dx, ref_dx))
        for t in out:
            t.fill_(7.0)
        ops.rows_resid_ln(a, pk, bias, resid, *out, 1e-6)
        n_bad[1] += int(not all(torch.equal(u, v) for u, v in zip(out, ref)))
    print(f'N = {N}, K = {K:4d}, M = {M}: {iters} launches each -- rows_lnbwd_t mismatches {n_bad[0]}, rows_resid_ln mismatches {n_bad[1]}', flush=True)
    bad += sum(n_bad)
print('OK: every launch bit-identical' if bad == 0 else f'FAILED: {bad} mismatching launches')
sys.exit(1 if bad else 0)
