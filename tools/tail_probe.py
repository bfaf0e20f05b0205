"""Does the nearly empty last round of tiles cost a whole round?  The residual GEMM (256 x 128 tiles, two workgroups per CU) and the
256 x 256 store kernel at M = 8.00 rounds of the chip and at the model's M = 8.07 rounds: ms per launch and ms per million rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402

ops, dev, BF = hip_ops.get(), 'cuda', torch.bfloat16
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


for name, N, K, epi in (('resid N=512 K=512', 512, 512, 2), ('resid N=512 K=1024', 512, 1024, 2), ('store N=512 K=1536', 512, 1536, 0), ('store N=1536 K=512', 1536, 512, 0)):
    for M in (262144, 264384, 266240, 294912):
        a = torch.randn(M, K, device=dev, generator=g).to(BF)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
        bias = torch.randn(N, device=dev, generator=g)
        if epi == 2:
            r, y = torch.randn(M, N, device=dev, generator=g), torch.empty(M, N, device=dev)
            fn = lambda: ops.gemm_nt(a, w, bias, 2, out_f=y, resid=r)
        else:
            o = torch.empty(M, N, device=dev, dtype=BF)
            fn = lambda: ops.gemm_nt(a, w, bias, 0, out_t=o)
        ms = timed(fn)
        print(f'{name:20s} M={M:7d} ({M / 256 / (512 if epi == 2 else 256) * (N / (128 if epi == 2 else 256)):.2f} rounds): {ms:.4f} ms  {ms / M * 1e6:.4f} ms per 1e6 rows', flush=True)
