"""Forward-only throughput (BASELINE.json configs[1]: full model, eval, no_grad, B=64 x 243 frames, bf16)."""
import os, sys, time
from functools import partial
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from motionbert_amd import DSTformer
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **bench.FULL).cuda().eval()
x, _ = bench.make_batch(B, 243, 17, 5, 'cuda')
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
fl = bench.model_flops_fwd(bench.FULL, 243) * B
print(f'peak HBM {torch.cuda.max_memory_allocated()/2**30:.1f} GiB; ', end='')
print(f'forward-only B={B}: {dt*1e3:.2f} ms, {B/dt:.1f} clips/s, {fl/dt/1e12:.1f} TFLOP/s = {fl/dt/2.5e15:.1%} of the bf16 MFMA peak')
