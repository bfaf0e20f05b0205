"""The LayerNorm-backward GEMM inside a Block, old and new: mbx_gemm_nt_lnbwd_t (256 x 128 tiles, row constants from the producers' row dots)
against mbx_rows_lnbwd_t (row owner, row means from the accumulators).  `python tools/lnbwd_time.py [clips]`  (library: MBX_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, N, dev, BF = clips * 243 * 17, 512, 'cuda', torch.bfloat16
ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=20):
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


print(f'# {os.path.basename(os.environ.get("MBX_LIB", "libmbx.so"))}  M = {M}, N = {N}', flush=True)
for name, K in (('d(qkv) -> dx', 1536), ('d(fc1) -> dx', 1024), ('K = 512', 512)):
    dy = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
    xhat = torch.randn(M, N, device=dev, generator=g).to(BF)
    rstd = torch.rand(M, device=dev, generator=g) + 0.5
    dres = torch.randn(M, N, device=dev, generator=g).to(BF)
    rowc = torch.rand(M, 4, device=dev, generator=g)
    out_old, out_new = torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, N, device=dev, dtype=BF)
    packed = ops.rows_n_pack(w)
    t_old = timed(lambda: ops.gemm_nt_lnbwd(dy, w, xhat, rowc, dres, None, None, out_old))
    t_new = timed(lambda: ops.rows_lnbwd_t(dy, packed, xhat, rstd, dres, out_new))
    t_plain = timed(lambda: ops.gemm_nt(dy, w, None, 0, out_t=out_old))
    fl = 2.0 * M * N * K
    print(f'{name:14s} K={K:5d}: tile kernel + row constants {t_old:.4f} ms = {fl / t_old / 1e9:5.0f} TF/s | row owner {t_new:.4f} ms = {fl / t_new / 1e9:5.0f} TF/s '
          f'({(t_new / t_old - 1) * 100:+.1f} %) | plain product on the tile kernel {t_plain:.4f} ms', flush=True)
