"""Diagnostics (-DMBX_TN_TRACE build): where a chunk of the weight-gradient GEMM (gemm_tn_pipe256_kernel) spends its cycles -- per workgroup,
for the first wave of the leading and of the trailing group: transposed reads + waits | first barrier | LDS-DMA issue + 16 MFMAs | second barrier.
    python tools/build_variants.py tntrace -DMBX_TN_TRACE
    MBX_LIB=tools/variants/libmbx_tntrace.so python tools/tn_trace.py [clips] [N] [K]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
M, dev, BF = clips * 243 * 17, 'cuda', torch.bfloat16
buf = torch.zeros(10 * 4096 + 64, dtype=torch.int64, device=dev)
os.environ['MBX_TRACE_BUF'] = hex(buf.data_ptr())
from motionbert_amd import hip_ops   # noqa: E402

ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
dy = (torch.randn(M, N, device=dev, generator=g) * 0.5).to(BF)
a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(BF)
dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
fn = lambda: ops.gemm_tn(dy, a, dw, db)
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
raw = buf.cpu().numpy()[:10 * 4096].reshape(-1, 2, 5)
raw = raw[raw[:, 0, 4] > 0]
print(f'# gemm_tn_pipe256_kernel dW [{N}, {K}] over M = {M}: {len(raw)} workgroups, launch {e0.elapsed_time(e1):.3f} ms (trace build), '
      f'{int(np.median(raw[:, 0, 4]))} chunks of 32 tokens per workgroup; shader cycles per chunk (median over workgroups)')
names = ['transposed reads + waits (+ bias dots)', 'first barrier', 'LDS-DMA issue + 16 MFMAs', 'second barrier']
for w, tag in ((0, 'leading group (wave 0)'), (1, 'trailing group (wave 4)')):
    per = raw[:, w, :4] / raw[:, w, 4:5]
    print(f'{tag}: ' + '   '.join(f'{nm} {np.median(per[:, k]):6.0f}' for k, nm in enumerate(names)) + f'   | chunk {np.median(per.sum(1)):6.0f}')
