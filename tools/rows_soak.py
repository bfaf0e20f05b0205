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
M, N, dev, BF = clips * 243 * 17, 512, 'cuda', torch.bfloat16
ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for K in (512, 1024, 1536):
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
    # spot check of the LAST rows against torch (the kernels address rows with 32-bit byte offsets: the highest ones are at the end)
    rows = torch.cat([torch.arange(M - 300, M, device=dev), torch.randint(0, M, (300,), device=dev, generator=g)])
    acc = a[rows].float() @ w.float().t()
    xh = xhat[rows].float()
    want_dx = dres[rows].float() + rstd[rows, None] * (acc - acc.mean(-1, keepdim=True) - xh * (acc * xh).mean(-1, keepdim=True))
    yy = resid[rows] + acc + bias
    mu = yy.mean(-1, keepdim=True)
    want_xh = (yy - mu) * torch.rsqrt(((yy - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
    e = [float((ref_dx[rows].float() - want_dx).norm() / want_dx.norm()), float((ref[0][rows] - yy).norm() / yy.norm()),
         float((ref[1][rows].float() - want_xh).norm() / want_xh.norm())]
    print(f'K = {K:4d}: last 300 + 300 random rows against torch: dx {e[0]:.2e}  y {e[1]:.2e}  xhat {e[2]:.2e}', flush=True)
    bad += int(e[0] > 6e-3 or e[1] > 1e-5 or e[2] > 6e-3)
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
    print(f'K = {K:4d}, M = {M}: {iters} launches each -- rows_lnbwd_t mismatches {n_bad[0]}, rows_resid_ln mismatches {n_bad[1]}', flush=True)
    bad += sum(n_bad)
print('OK: every launch bit-identical' if bad == 0 else f'FAILED: {bad} mismatching launches')
sys.exit(1 if bad else 0)
