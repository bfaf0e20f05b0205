import os, sys, torch
sys.path.insert(0, '.')
from motionbert_amd import hip_ops
ops = hip_ops.get()
M, dev, BF = 64 * 243 * 17, 'cuda', torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in (('lite dW_qkv', 768, 256), ('lite dW_proj', 256, 256), ('lite dW_fc1', 1024, 256), ('lite dW_fc2', 256, 1024), ('full dW_qkv', 1536, 512)):
    dy = (torch.randn(M, N, device=dev, generator=g) * 0.5).to(BF)
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(BF)
    dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    fn = lambda: ops.gemm_tn(dy, a, dw, db)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ref = dy.float().t() @ a.float()
    err = float((dw - ref).norm() / ref.norm()); errb = float((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm())
    print(f'{name:13s} [{N}, {K}]: {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:6.0f} TF/s  err {err:.1e} / {errb:.1e}', flush=True)
