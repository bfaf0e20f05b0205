"""Evidence for the ceiling argument: engine clock and package power, sampled from amd-smi / rocm-smi every
~0.15 s, WHILE (a) nothing runs, (b) the MFMA-only probe runs (mbx_mfma_probe: nothing but v_mfma_f32_32x32x16_bf16, random
operands), (c) the bench line's training step runs (64 clips x 243 frames, bf16, fwd + loss + bwd + AdamW), (d) the no-grad forward
runs, (e) an HBM-bound kernel (LayerNorm forward) runs; next to each the in-kernel clock (shader cycles / real time) where a kernel
reports it.   python tools/clock_power.py > profiles/r05_clock_power.txt"""
import os
import re
import subprocess
import sys
import threading
import time
from functools import partial

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from motionbert_amd import hip_ops
from motionbert_amd.model import DSTformer
from motionbert_amd.train import FlatAdamW, pose_loss as fused_pose_loss

ops = hip_ops.get()
dev = torch.device('cuda', 0)
samples, stop = [], False


def read_smi():
    """(sclk MHz, power W) from whichever tool answers"""
    try:
        o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5).stdout
        hdr, row = [l.split(',') for l in o.strip().splitlines()[:2]]
        d = dict(zip(hdr, row))
        sclk = next((re.search(r'(\d+)\s*Mhz', v, re.I) for k, v in d.items() if 'sclk' in k.lower()), None)
        pw = next((v for k, v in d.items() if 'power' in k.lower() and re.match(r'^[\d.]+$', v)), None)
        return (float(sclk.group(1)) if sclk else None, float(pw) if pw else None, None)
    except Exception as e:
        return (None, None, repr(e)[:80])


def sampler():
    while not stop:
        samples.append((time.time(),) + read_smi())
        time.sleep(0.1)


def phase(name, fn, secs, extra=None):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    info = None
    while time.time() - t0 < secs:
        info = fn()
        torch.cuda.synchronize()
        n += 1
    t1 = time.time()
    mine = [s for s in samples if t0 + 0.5 <= s[0] <= t1]      # skip the first half second (power management settling)
    clk = [s[1] for s in mine if s[1]]
    pw = [s[2] for s in mine if s[2]]
    line = f'{name:58s}: {n:5d} calls, {(t1 - t0) / max(n, 1) * 1e3:9.3f} ms each | smi sclk {min(clk):.0f}-{max(clk):.0f} MHz (median {sorted(clk)[len(clk) // 2]:.0f}), ' \
           f'power {min(pw):.0f}-{max(pw):.0f} W (median {sorted(pw)[len(pw) // 2]:.0f}), {len(mine)} samples' if clk and pw else f'{name:58s}: {n} calls; no smi samples ({mine[:1]})'
    if extra and info is not None:
        line += ' | ' + extra(info)
    print(line, flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
print('# tools/clock_power.py on', torch.cuda.get_device_name(0), '-- smi sampled every ~0.15 s during each phase (first 0.5 s of a phase dropped)')
phase('idle', lambda: time.sleep(0.2), 2.0)
phase('MFMA-only probe, 1 wave / SIMD (mbx_mfma_probe, random operands)', lambda: ops.mfma_probe(0.5), 4.0,
      lambda r: f"in-kernel: {r['tflops']:.0f} TFLOP/s at {r['clock_ghz']:.3f} GHz")
phase('MFMA-only probe, 2 workgroups / CU', lambda: ops.mfma_probe(0.5, wgs_per_cu=2), 3.0,
      lambda r: f"in-kernel: {r['tflops']:.0f} TFLOP/s at {r['clock_ghz']:.3f} GHz")

FULL = dict(dim_in=3, dim_out=3, dim_feat=512, dim_rep=512, depth=5, num_heads=8, mlp_ratio=2, num_joints=17, maxlen=243)
torch.manual_seed(0)
model = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **FULL).to(dev)
model.precision = 'bf16'
model.train()
opt = FlatAdamW(model, lr=2e-4, weight_decay=0.01)
g = torch.Generator(device=dev).manual_seed(0)
B, T, J = 64, 243, 17
x = torch.rand(B, T, J, 3, device=dev, generator=g) * 2 - 1
gt = torch.randn(B, T, J, 3, device=dev, generator=g) * 0.3


def step():
    opt.zero_grad(set_to_none=True)
    total, _ = fused_pose_loss(model(x), gt, 0.5, 20.0)
    total.backward()
    opt.step()


for _ in range(3):
    step()
phase('training step (bench line: 64 clips, fwd + loss + bwd + AdamW)', step, 6.0)
model.eval()


def fwd():
    with torch.no_grad():
        model(x)


phase('no-grad forward (64 clips)', fwd, 4.0)
M = B * T * J
xx, y = torch.randn(M, 512, device=dev), torch.empty(M, 512, device=dev, dtype=torch.bfloat16)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
phase('LayerNorm forward (HBM-bound)', lambda: [ops.layernorm_fwd(xx, None, None, 1e-6, y, mean, rstd) for _ in range(50)] and None, 3.0)
phase('MFMA-only probe again (after the model phases)', lambda: ops.mfma_probe(0.5), 3.0,
      lambda r: f"in-kernel: {r['tflops']:.0f} TFLOP/s at {r['clock_ghz']:.3f} GHz")
stop = True
