"""Diagnostics (-DMBX_MLP_TRACE build): where a 128-row tile of the fused proj + MLP kernel spends its time IN SITU -- 15 time stamps
per workgroup (s_memrealtime, 10 ns ticks), all workgroups of one launch (-DMBX_MLP_TRACE=2: also the four stages of one chunk).
    python tools/build_variants.py mlptrace -DMBX_MLP_TRACE
    MBX_LIB=tools/variants/libmbx_mlptrace.so python tools/mlp_trace.py [clips] [proj=1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
proj = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
C, hidden, M, dev, BF = 512, 1024, clips * 243 * 17, 'cuda', torch.bfloat16
tiles = (M + 127) // 128
buf = torch.zeros(24 * tiles + 64, dtype=torch.int64, device=dev)
os.environ['MBX_TRACE_BUF'] = hex(buf.data_ptr())
from motionbert_amd import hip_ops   # noqa: E402

ops = hip_ops.get()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, C, device=dev, generator=g)
a = x.to(BF)
w1 = (torch.randn(hidden, C, device=dev, generator=g) * 0.05).to(BF)
w2 = (torch.randn(C, hidden, device=dev, generator=g) * 0.05).to(BF)
wp = (torch.randn(C, C, device=dev, generator=g) * 0.05).to(BF)
b1, b2, bp = torch.randn(hidden, device=dev, generator=g), torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
rsum = w1.float().sum(1)
y = torch.empty(M, C, device=dev)
if proj:
    pk = ops.proj_mlp_pack_weights(wp, w1, w2)
    fn = lambda: ops.proj_mlp_fused_fwd(a, pk, bp, b1, b2, rsum, x, y, 1e-6)
else:
    pk = ops.mlp_pack_weights(w1, w2)
    fn = lambda: ops.mlp_fused_fwd(None, 1, pk, b1, b2, rsum, x, y, None, 1e-6, None, None)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
t = buf.cpu().numpy()
raw = t[:24 * tiles].reshape(tiles, 24)
st = raw[:, :16].astype(np.float64)
cyc0, cyc1 = raw[:, 22].astype(np.float64), raw[:, 23].astype(np.float64)
if not proj:
    st[:, 7] = st[:, 6]
for k in (1, 2, 3):      # the pipelined prologue (round 5) is one phase: stamps 1..3 are not written
    st[:, k] = np.where(st[:, k] == 0, st[:, k - 1], st[:, k])
t0 = st[:, 0].min()
us = (st[:, :15] - t0) / 100.0          # 100 MHz ticks -> us since the first workgroup's entry
names = ['o -> X fragments (DMA + wait + reads)', 'barrier', 'residual half 0 -> acc2', 'residual half 1 -> acc2', 'fp32 -> operand / barrier',
         'first weight stages landed', 'proj stages', 'y1 -> operand + statistics', 'A(0) + gelu(0)', 'chunk loop', 'B(n-1)',
         'drain + barrier', 'epilogue (image + stores issued)', 'stores acknowledged']
dur = np.diff(us, axis=1)
total = us[:, 14] - us[:, 0]
order = np.argsort(us[:, 0])
rounds = [('first 256 workgroups', order[:256]), ('workgroups 2048..4095', order[2048:4096]), ('last 1024 workgroups', order[-1024:]), ('all', order)]
print(f'# mlp_fused_kernel<512, {"true" if proj else "false"}> at {clips} clips: {tiles} tiles, launch {e0.elapsed_time(e1):.3f} ms (trace build), '
      f'{e0.elapsed_time(e1) * 1e3 / (tiles / 256):.1f} us per tile round')
print(f'# effective shader clock inside a tile: {np.median((cyc1 - cyc0) / (total * 1e-6)) / 1e9:.3f} GHz (median over workgroups)')
print(f'{"phase (median us per tile)":46s}' + ''.join(f'{n:>24s}' for n, _ in rounds))
for k, nm in enumerate(names):
    print(f'{nm:46s}' + ''.join(f'{np.median(dur[idx, k]):24.2f}' for _, idx in rounds))
print(f'{"whole tile":46s}' + ''.join(f'{np.median(total[idx]):24.2f}' for _, idx in rounds))
print(f'{"whole tile, 10th / 90th percentile":46s}' + ''.join(f'{np.percentile(total[idx], 10):11.1f} /{np.percentile(total[idx], 90):10.1f} ' for _, idx in rounds))
# how synchronised are the CUs?  spread of tile start times inside successive groups of 256 workgroups (one per CU)
starts = np.sort(us[:, 0])
for r in (0, 1, 4, 8, 16, 24, 31):
    seg = starts[256 * r:256 * (r + 1)]
    if len(seg):
        print(f'# start times of workgroups {256 * r}..{256 * r + len(seg) - 1}: {seg.min():.1f} .. {seg.max():.1f} us (spread {seg.max() - seg.min():.1f}, std {seg.std():.1f})')
# memory phases in flight at the same time: fraction of CUs inside (prologue loads | epilogue stores) over time
grid = np.linspace(0, us[:, 14].max(), 2000)
inpro = ((us[:, 0][None, :] <= grid[:, None]) & (grid[:, None] < us[:, 5][None, :])).sum(1)
inepi = ((us[:, 12][None, :] <= grid[:, None]) & (grid[:, None] < us[:, 14][None, :])).sum(1)
print(f'# workgroups inside the load phases at a time: mean {inpro.mean():.1f}, max {inpro.max()}; inside the store phase: mean {inepi.mean():.1f}, max {inepi.max()} (of 256 resident)')
if raw[:, 16].min() > 0:      # -DMBX_MLP_TRACE=2: the four stages of chunk 8
    sd = np.diff(raw[:, 16:21].astype(np.float64), axis=1) / 100.0
    print('# chunk 8, median us per stage: ' + '  '.join(f'{n} {np.median(sd[:, k]):.2f}' for k, n in enumerate(('A0 (fc1)', 'A1 (fc1)', 'B0 (fc2 + GELU)', 'B1 (fc2 + GELU)'))))
xcc = (raw[:, 15] >> 32) & 0xf
print('# median tile time by XCC: ' + ' '.join(f'{int(c)}:{np.median(total[xcc == c]):.1f}' for c in np.unique(xcc)))
