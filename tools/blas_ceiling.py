"""Diagnostics: what the vendor GEMM (hipBLASLt through torch) reaches on the model's shapes -- a ceiling to compare
the hand-written kernels against, not a code path of the product."""
import torch, time
M = 64 * 243 * 17
bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, K in [('qkv', 1536, 512), ('proj', 512, 512), ('fc1', 1024, 512), ('fc2', 512, 1024), ('dX_qkv', 512, 1536)]:
    a = torch.randn(M, K, device='cuda', dtype=bf); w = torch.randn(N, K, device='cuda', dtype=bf) * 0.05; b = torch.randn(N, device='cuda', dtype=bf)
    out = torch.empty(M, N, device='cuda', dtype=bf)
    ms = t(lambda: torch.matmul(a, w.t(), out=out))
    ms2 = t(lambda: torch.nn.functional.linear(a, w, b))
    print(f'NT {name:7s} N={N} K={K}: matmul {ms:.4f} ms {2*M*N*K/ms/1e9:7.1f} TF/s | linear+bias {ms2:.4f} ms {2*M*N*K/ms2/1e9:7.1f} TF/s', flush=True)
for name, N, K in [('dW_qkv', 1536, 512), ('dW_fc1', 1024, 512), ('dW_proj', 512, 512)]:
    dy = torch.randn(M, N, device='cuda', dtype=bf); a = torch.randn(M, K, device='cuda', dtype=bf)
    ms = t(lambda: torch.matmul(dy.t(), a))
    print(f'TN {name:7s} N={N} K={K}: matmul {ms:.4f} ms {2*M*N*K/ms/1e9:7.1f} TF/s', flush=True)
