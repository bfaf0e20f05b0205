"""MotionBERT-Lite (MB_lite.yaml: dim_feat 256, mlp_ratio 4) training step and no-grad forward alone -- the command to put under
rocprofv3 for the C = 256 / head-dim-32 kernels.   python tools/lite_bench.py [clips] [steps]"""
import os
import sys
import time
from functools import partial

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from motionbert_amd import DSTformer   # noqa: E402
from motionbert_amd.train import FlatAdamW, pose_loss   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(1)
m = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **bench.LITE).cuda()
opt = FlatAdamW(m, lr=2e-4, weight_decay=0.01)
x, gt = bench.make_batch(B, 243, 17, 5, 'cuda')


def step():
    opt.zero_grad(set_to_none=True)
    total, _ = pose_loss(m(x), gt, bench.LAMBDA_SCALE, bench.LAMBDA_VELOCITY)
    total.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
m.eval()
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m(x)
    torch.cuda.synchronize()
fd = (time.perf_counter() - t0) / steps
fl = bench.model_flops_fwd(bench.LITE, 243) * B
print(f'Lite B={B}: train {dt * 1e3:.2f} ms/step = {B / dt:.0f} clips/s ({3 * fl / dt / 1e12:.0f} TFLOP/s); forward only {fd * 1e3:.2f} ms = {B / fd:.0f} clips/s', flush=True)
