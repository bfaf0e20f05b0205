"""Diagnostics: per-k-tile cycle stamps of one wave of gemm_nt_pipe (MBX_TRACE_BUF)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
buf = torch.zeros(4096, dtype=torch.int64, device='cuda')
os.environ['MBX_TRACE_BUF'] = hex(buf.data_ptr())
from motionbert_amd import hip_ops
from motionbert_amd.engine import EPI_STORE, EPI_RESID
ops = hip_ops.get()
M, N, K = 64 * 243 * 17, int(sys.argv[1]) if len(sys.argv) > 1 else 1536, int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else 'store'      # store | resid | lnbwd (the 256x128 kernel's epilogues)
a = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
outf, resid = torch.empty(M, N, device='cuda'), torch.randn(M, N, device='cuda')
rowc = torch.rand(M, 4, device='cuda')
for it in range(3):
    if mode == 'resid':
        ops.gemm_nt(a, w, None, EPI_RESID, out_f=outf, resid=resid)
    elif mode == 'lnbwd':
        ops.gemm_nt_lnbwd(a, w, out, rowc, resid, None, outf, torch.empty_like(out))
    else:
        ops.gemm_nt(a, w, None, EPI_STORE, out_t=out)
torch.cuda.synchronize()
t = buf.cpu().tolist()
nk = K // 32
t0 = t[0]
print(f'N={N} K={K} {mode}'); print('k-tile: wait(vmcnt)  barrier  issue  compute   [cycles]')
for kt in range(nk):
    b = 1 + kt * 4
    prev = t[b - 1] if kt else t0
    print(f'{kt:3d}: {t[b]-prev:8d} {t[b+1]-t[b]:8d} {t[b+2]-t[b+1]:6d} {t[b+3]-t[b+2]:8d}')
e = 1 + nk * 4
print(f'epilogue issue {t[e]-t[e-1]} cycles, store drain {t[e+1]-t[e]} cycles, total tile {t[e+1]-t0} cycles')
