"""Diagnostics (-DMBX_ROWS_TRACE build): where a 128-row tile of the row-owner qkv kernel (mbx_rows_gemm_nk_ln: LayerNorm + qkv from the
fp32 rows, N = 1536, K = 512) spends its time in situ -- six time stamps per workgroup, all workgroups of one launch.
    python tools/build_variants.py rowstrace -DMBX_ROWS_TRACE
    MBX_LIB=tools/variants/libmbx_rowstrace.so python tools/rows_trace.py [clips]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, K, M, dev, BF = 1536, 512, clips * 243 * 17, 'cuda', torch.bfloat16
tiles = (M + 127) // 128
buf = torch.zeros(8 * (tiles + 8 * 512) + 64, dtype=torch.int64, device=dev)
os.environ['MBX_TRACE_BUF'] = hex(buf.data_ptr())
from motionbert_amd import hip_ops   # noqa: E402

ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
bias = torch.randn(N, device=dev, generator=g)
rsum = w.float().sum(1)
packed = ops.rows_pack_nk(w)
out = torch.empty(M, N, device=dev, dtype=BF)
fn = lambda: ops.rows_gemm_nk_ln(x, packed, bias, rsum, 1e-6, out)
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
t = buf.cpu().numpy()
raw = t[:8 * (tiles + 8 * 512)].reshape(-1, 8)
raw = raw[raw[:, 0] > 0]
full = raw[:(tiles // 512) * 512] if tiles >= 512 else raw
st = full[:, :6].astype(np.float64)
us = (st - st[:, 0].min()) / 100.0
names = ['prologue (fp32 rows -> operand + statistics, first stages landed)', 'chunk 0 (two 32-column tiles)', 'chunks 1 .. n/2', 'chunks n/2 .. n', 'last epilogue + stores acknowledged']
dur = np.diff(us, axis=1)
total = us[:, 5] - us[:, 0]
print(f'# rows_nk_kernel<512, LN, from fp32 rows> at {clips} clips: {tiles} tiles ({len(raw)} workgroups), launch {e0.elapsed_time(e1):.3f} ms (trace build) = '
      f'{2.0 * M * N * K / e0.elapsed_time(e1) / 1e9:.0f} TFLOP/s; two workgroups per CU')
print(f'# effective shader clock: {np.median(full[:, 7] / (total * 1e-6)) / 1e9:.3f} GHz')
for k, nm in enumerate(names):
    print(f'{nm:72s} {np.median(dur[:, k]):8.2f} us (10th / 90th percentile {np.percentile(dur[:, k], 10):.2f} / {np.percentile(dur[:, k], 90):.2f})')
print(f'{"whole tile":72s} {np.median(total):8.2f} us (10th / 90th percentile {np.percentile(total, 10):.2f} / {np.percentile(total, 90):.2f})')
nch = N // 64
per_chunk = np.median(dur[:, 2] + dur[:, 3]) / (nch - 1)
print(f'# steady state: {per_chunk:.2f} us per chunk of 64 MFMA slots = {per_chunk / 64 * 1e3:.1f} ns per slot; one v_mfma_f32_32x32x16_bf16 = 32 cycles')
