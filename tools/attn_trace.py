"""Diagnostics (-DMBX_ATTN_TRACE build): where a problem of the temporal attention backward (attn_bwd_fused_kernel: one (clip, joint, head)
sequence of 243 frames per workgroup of sixteen waves) spends its time in situ -- eight time stamps per workgroup, all of one launch.
    python tools/build_variants.py attntrace -DMBX_ATTN_TRACE
    MBX_LIB=tools/variants/libmbx_attntrace.so python tools/attn_trace.py [clips]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, J, H, hd, dev = 243, 17, 8, 64, 'cuda'
C, M, nprob = H * hd, B * T * J, B * J * H
buf = torch.zeros(25 * nprob + 64, dtype=torch.int64, device=dev)
os.environ['MBX_TRACE_BUF'] = hex(buf.data_ptr())
from motionbert_amd import hip_ops   # noqa: E402
from motionbert_amd.engine import MODE_TEMPORAL   # noqa: E402

ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(M, 3 * C, device=dev, generator=g).to(torch.bfloat16)
d_o = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
o, lse = torch.empty(M, C, device=dev, dtype=torch.bfloat16), torch.empty(M, H, device=dev)
dqkv = torch.empty_like(qkv)
ops.attn_fwd(qkv, o, lse, B, T, J, H, hd ** -0.5, MODE_TEMPORAL)
fn = lambda: ops.attn_bwd(qkv, o, d_o, lse, dqkv, B, T, J, H, hd ** -0.5, MODE_TEMPORAL)
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
raw = buf.cpu().numpy()[:9 * nprob].reshape(-1, 9)
raw = raw[raw[:, 0] > 0]
us = (raw[:, :8].astype(np.float64) - raw[:, 0].min()) / 100.0
order = np.argsort(us[:, 0])
steady = us[order][512:len(us) - 512]
names = ['fill: loads issued, landed, written to the LDS tiles', 'statistics + first barrier', 'compute (dQ | dK, dV)', 'second barrier (slowest wave)',
         'gradients staged over the tiles + third barrier', 'copy-out: LDS reads + stores issued', 'stores acknowledged']
dur = np.diff(steady, axis=1)
total = steady[:, 7] - steady[:, 0]
print(f'# attn_bwd_fused_kernel<64> at {B} clips: {nprob} problems, launch {e0.elapsed_time(e1):.3f} ms (trace build); steady-state problems: {len(steady)}')
for k, nm in enumerate(names):
    print(f'{nm:58s} {np.median(dur[:, k]):7.2f} us (10th / 90th percentile {np.percentile(dur[:, k], 10):.2f} / {np.percentile(dur[:, k], 90):.2f})')
print(f'{"whole problem":58s} {np.median(total):7.2f} us (10th / 90th percentile {np.percentile(total, 10):.2f} / {np.percentile(total, 90):.2f})')
pw = buf.cpu().numpy()[9 * nprob:25 * nprob].reshape(-1, 16).astype(np.float64)
ok = buf.cpu().numpy()[:9 * nprob].reshape(-1, 9)[:, 0] > 0
t2 = buf.cpu().numpy()[:9 * nprob].reshape(-1, 9)[:, 2].astype(np.float64)
rel = (pw[ok] - t2[ok][:, None]) / 100.0
print('# end of the compute phase per wave, us after the first barrier (median over all problems): ' + ' '.join(f'{np.median(rel[:, w]):.1f}' for w in range(16)))
print(f'# launch / (problems / 256 CUs) = {e0.elapsed_time(e1) * 1e3 / (nprob / 256):.2f} us per problem and CU')
