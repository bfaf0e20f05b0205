"""ms per launch of mbx_mlp_fused_fwd (raw operand in, y + bf16 y + statistics out) for the library in MBX_LIB: `python tools/mlp_time.py [clips] [C]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import hip_ops   # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
hidden, M, dev, BF = 1024, clips * 243 * 17, 'cuda', torch.bfloat16
ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, C, device=dev, generator=g)
a = x.to(BF)
w1 = (torch.randn(hidden, C, device=dev, generator=g) * 0.05).to(BF)
w2 = (torch.randn(C, hidden, device=dev, generator=g) * 0.05).to(BF)
b1, b2 = torch.randn(hidden, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
rsum, packed = w1.float().sum(1), ops.mlp_pack_weights(w1, w2)
y, yt, mean, rstd = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev, dtype=BF), torch.empty(M, device=dev), torch.empty(M, device=dev)
from_x = os.environ.get('MLP_FROM_X', '1') == '1'      # the product's form: the operand is made from the fp32 rows in the kernel
fn = lambda: ops.mlp_fused_fwd(None if from_x else a, 1, packed, b1, b2, rsum, x, y, yt, 1e-6, mean, rstd)
if os.environ.get('MLP_PROJ', '0') == '1':      # proj + residual in front, same kernel (vs the residual GEMM + the fused MLP as two launches)
    wp = (torch.randn(C, C, device=dev, generator=g) * 0.05).to(BF)
    bp = torch.randn(C, device=dev, generator=g)
    pp = ops.proj_mlp_pack_weights(wp, w1, w2)
    y1 = torch.empty(M, C, device=dev)
    two = lambda: (ops.gemm_nt(a, wp, bp, 2, out_f=y1, resid=x), ops.mlp_fused_fwd(None, 1, packed, b1, b2, rsum, y1, y, None, 1e-6, None, None))
    for _ in range(3):
        two()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        two()
    t1.record()
    torch.cuda.synchronize()
    print(f'two launches (proj + residual GEMM, fused MLP): {t0.elapsed_time(t1) / 10:.3f} ms', flush=True)
    fn = lambda: ops.proj_mlp_fused_fwd(a, pp, bp, b1, b2, rsum, x, y, 1e-6)
    from_x = 'proj'
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
tiles = (M + 127) // 128
print(f'{os.path.basename(os.environ.get("MBX_LIB", "libmbx.so")):28s} clips={clips} C={C} {("proj + MLP" if from_x == "proj" else "from_x") if from_x else "bf16 operand"}: {ms:.3f} ms = {4.0 * M * C * hidden / ms / 1e9:.0f} TF/s, '
      f'{ms * 1e3 / (tiles / 256):.1f} us per tile round', flush=True)
