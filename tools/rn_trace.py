"""Diagnostics (-DMBX_RN_TRACE build): where a 128-row tile of the row-owner LayerNorm-backward GEMM (mbx_rows_lnbwd_t) spends its time in
situ -- eleven time stamps per workgroup, all workgroups of one launch.
    python tools/build_variants.py rntrace -DMBX_RN_TRACE
    MBX_LIB=tools/variants/libmbx_rntrace.so python tools/rn_trace.py [clips] [K]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N, M, dev, BF = 512, clips * 243 * 17, 'cuda', torch.bfloat16
tiles = (M + 127) // 128
buf = torch.zeros(12 * tiles + 64, dtype=torch.int64, device=dev)
os.environ['MBX_TRACE_BUF'] = hex(buf.data_ptr())
from motionbert_amd import hip_ops   # noqa: E402

ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
dy = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(BF)
w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(BF)
xhat = torch.randn(M, N, device=dev, generator=g).to(BF)
rstd = torch.rand(M, device=dev, generator=g) + 0.5
dres = torch.randn(M, N, device=dev, generator=g).to(BF)
out = torch.empty(M, N, device=dev, dtype=BF)
packed = ops.rows_n_pack(w)
fn = lambda: ops.rows_lnbwd_t(dy, packed, xhat, rstd, dres, out)
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
raw = buf.cpu().numpy()[:12 * tiles].reshape(-1, 12)
raw = raw[raw[:, 0] > 0]
st = raw[:, :11].astype(np.float64)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
order = np.argsort(us[:, 0])
steady = us[order][256:len(us) - 256] if len(us) > 1024 else us       # not the first and not the last round
names = ['prologue (first stage and tokens landed)', 'loop', 'drain + barrier', 'xhat registers -> LDS', 'pass 1 (row means) + row constants',
         'pass 2 quarter 0 (dx, stores issued)', 'pass 2 quarter 1', 'pass 2 quarter 2', 'pass 2 quarter 3', 'stores acknowledged']
dur = np.diff(steady, axis=1)
total = steady[:, 10] - steady[:, 0]
print(f'# rows_n_lnbwd_kernel at {clips} clips, K = {K}: {tiles} tiles ({len(raw)} workgroups), launch {e0.elapsed_time(e1):.3f} ms (trace build); steady-state tiles: {len(steady)}')
for k, nm in enumerate(names):
    print(f'{nm:52s} {np.median(dur[:, k]):8.2f} us (10th / 90th percentile {np.percentile(dur[:, k], 10):.2f} / {np.percentile(dur[:, k], 90):.2f})')
print(f'{"whole tile":52s} {np.median(total):8.2f} us (10th / 90th percentile {np.percentile(total, 10):.2f} / {np.percentile(total, 90):.2f})')
# gaps between consecutive workgroups on the same CU (hardware id in column 11)
hw = raw[:, 11]
gaps = []
for h in np.unique(hw):
    rows_ = us[hw == h]
    rows_ = rows_[np.argsort(rows_[:, 0])]
    if len(rows_) > 1:
        gaps += list(rows_[1:, 0] - rows_[:-1, 10])
if gaps:
    print(f'# gap between a workgroup\'s last store acknowledged and the next workgroup\'s entry on the same CU: median {np.median(gaps):.2f} us '
          f'(10th / 90th percentile {np.percentile(gaps, 10):.2f} / {np.percentile(gaps, 90):.2f}); workgroups per CU: {len(raw) / len(np.unique(hw)):.2f}')
