"""Soak test of the no-grad path: the same forward N times (two streams, row-owner qkv, proj + MLP kernel); every output must be
bit-identical to the first -- a counted-wait that is one too weak shows up here as a rare mismatch.  python tools/nograd_soak.py [iters] [clips]"""
import os
import sys
from functools import partial

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from motionbert_amd import DSTformer   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for cfg_name, cfg in (('full', bench.FULL), ('lite', bench.LITE)):
    torch.manual_seed(0)
    m = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **cfg).cuda().eval()
    x, _ = bench.make_batch(B, 243, 17, 5, 'cuda')
    bad = 0
    with torch.no_grad():
        ref = m(x).clone()
        assert torch.isfinite(ref).all()
        for i in range(iters):
            out = m(x)
            if not torch.equal(out, ref):
                bad += 1
                print(f'{cfg_name}: iteration {i}: {int((out != ref).sum())} elements differ, max {float((out - ref).abs().max()):.3e}', flush=True)
    print(f'{cfg_name} B={B}: {iters} forwards, {bad} differ from the first', flush=True)
