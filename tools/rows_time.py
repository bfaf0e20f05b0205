"""Row-owner GEMM (mbx_rows_gemm_nk) against the tile kernels at the model's shapes: error vs an fp32 product of the same bf16 operands,
ms per launch, TF/s.  `python tools/rows_time.py [clips]`  (library: MBX_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, dev, BF = clips * 243 * 17, 'cuda', torch.bfloat16
ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f'# {os.path.basename(os.environ.get("MBX_LIB", "libmbx.so"))}  M = {M}', flush=True)
for name, N, K, ln in (('qkv', 1536, 512, False), ('qkv raw-LN', 1536, 512, True), ('qkv from x', 1536, 512, 'x'), ('fc1-shape', 1024, 512, False), ('proj-shape', 512, 512, False),
                       ('lite qkv', 768, 256, False)):
    a = torch.randn(M, K, device=dev, generator=g).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
    bias = torch.randn(N, device=dev, generator=g)
    rsum = w.float().sum(1)
    mean, rstd = torch.randn(M, device=dev, generator=g) * 0.1, torch.rand(M, device=dev, generator=g) + 0.5
    out, ref_out = torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, N, device=dev, dtype=BF)
    packed = ops.rows_pack_nk(w)
    if ln == 'x':
        x = torch.randn(M, K, device=dev, generator=g) * 0.7 + 0.3
        a = x.to(BF)
        mean, var = x.mean(1), x.var(1, unbiased=False)
        rstd = torch.rsqrt(var + 1e-6)
        fn = lambda: ops.rows_gemm_nk_ln(x, packed, bias, rsum, 1e-6, out)

        xhat, mu_, rs_ = torch.empty(M, K, device=dev, dtype=BF), torch.empty(M, device=dev), torch.empty(M, device=dev)

        def old():      # what the row kernel replaces: a LayerNorm pass + the tile GEMM on its output (unit gamma: the folded form)
            ops.layernorm_fwd(x, None, None, 1e-6, xhat, mu_, rs_)
            ops.gemm_nt(xhat, w, bias, 0, out_t=ref_out)
    elif ln:
        fn = lambda: ops.rows_gemm_nk(a, packed, bias, out, rsum, mean, rstd)
        old = lambda: ops.gemm_nt(a, w, bias, 0, out_t=ref_out)      # (the tile kernel has no raw-operand epilogue any more: plain product for the time)
    else:
        fn = lambda: ops.rows_gemm_nk(a, packed, bias, out)
        old = lambda: ops.gemm_nt(a, w, bias, 0, out_t=ref_out)
    fn()
    torch.cuda.synchronize()
    rows = slice(0, 4096)
    acc = a[rows].float() @ w.float().t()
    ref = (rstd[rows, None] * (acc - mean[rows, None] * rsum[None]) + bias) if ln else acc + bias
    err = ((out[rows].float() - ref).norm() / ref.norm()).item()
    tail = a[-300:].float() @ w.float().t()
    tref = (rstd[-300:, None] * (tail - mean[-300:, None] * rsum[None]) + bias) if ln else tail + bias
    terr = ((out[-300:].float() - tref).norm() / tref.norm()).item()
    ms = timed(fn)
    try:
        ms_old = timed(old)
        old()
        same = (out.float() - ref_out.float()).abs().max().item()
    except Exception as e:   # noqa: BLE001
        ms_old, same = float('nan'), float('nan')
    print(f'{name:12s} N={N:5d} K={K:4d}: rows {ms:.4f} ms = {2.0 * M * N * K / ms / 1e9:6.0f} TF/s   tile kernel {ms_old:.4f} ms   '
          f'rel err {err:.2e} (tail rows {terr:.2e})  max |rows - tile| {same:.3g}', flush=True)
