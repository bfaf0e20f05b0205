"""The forward residual GEMM + the LayerNorm that follows it, old and new: mbx_gemm_nt (residual epilogue, 256 x 128 tiles) + mbx_layernorm_fwd
against mbx_rows_resid_ln (row owner: the statistics from its own registers).  `python tools/residln_time.py [clips]`  (library: MBX_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402
from motionbert_amd.engine import EPI_RESID   # noqa: E402

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
for name, K in (('proj + residual -> LayerNorm', 512), ('fc2 + residual -> LayerNorm', 1024)):
    a = (torch.randn(M, K, device=dev, generator=g) * 0.7).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
    bias = torch.randn(N, device=dev, generator=g) * 0.3
    resid = torch.randn(M, N, device=dev, generator=g)
    y, xn, mean, rstd = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, device=dev), torch.empty(M, device=dev)
    packed = ops.rows_n_pack(w)
    t_gemm = timed(lambda: ops.gemm_nt(a, w, bias, EPI_RESID, resid=resid, out_f=y))
    t_ln = timed(lambda: ops.layernorm_fwd(y, None, None, 1e-6, xn, mean, rstd))
    t_new = timed(lambda: ops.rows_resid_ln(a, packed, bias, resid, y, xn, mean, rstd, 1e-6))
    gb = (M * K * 2 + M * N * (4 + 4 + 2)) / 1e9
    print(f'{name:30s} K={K:5d}: tile kernel {t_gemm:.4f} + LayerNorm {t_ln:.4f} = {t_gemm + t_ln:.4f} ms | row owner {t_new:.4f} ms '
          f'({(t_new / (t_gemm + t_ln) - 1) * 100:+.1f} %; {gb:.2f} GB algorithmic -> {gb / t_new:.2f} TB/s)', flush=True)
