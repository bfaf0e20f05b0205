"""Diagnostics: engine clock and socket power while the QKV GEMM / a LayerNorm loop runs (is the MFMA peak of the
datasheet, quoted at 2.4 GHz, reachable under this load, or does the chip sit at a lower DVFS point?)."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionbert_amd import hip_ops
from motionbert_amd.engine import EPI_STORE
ops = hip_ops.get()
M = 64 * 243 * 17
bf = torch.bfloat16
a, w, b = torch.randn(M, 512, device='cuda').to(bf), (torch.randn(1536, 512, device='cuda') * 0.05).to(bf), torch.randn(1536, device='cuda')
out = torch.empty(M, 1536, device='cuda', dtype=bf)
x = torch.randn(M, 512, device='cuda'); g = torch.ones(512, device='cuda'); y = torch.empty(M, 512, device='cuda', dtype=bf)
mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), o.strip().replace('\n', ' | ')))
        except Exception as e:
            samples.append((time.time(), repr(e)))
        time.sleep(0.2)
def phase(name, fn, secs):
    global samples
    samples = []
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    print(f'== {name}: {n} launches in {dt:.2f}s = {dt / n * 1e3:.4f} ms each')
    for t, s in samples[-3:]:
        print('   ', s[:600])
th = threading.Thread(target=sampler, daemon=True); th.start()
phase('idle', lambda: None, 1.5)
phase('gemm_nt qkv', lambda: ops.gemm_nt(a, w, b, EPI_STORE, out_t=out), 4)
phase('layernorm_fwd', lambda: ops.layernorm_fwd(x, g, g, 1e-6, y, mean, rstd), 3)
stop = True
