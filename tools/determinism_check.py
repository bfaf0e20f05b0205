"""Diagnostics: run the same full-size forward+backward several times and report which parameter gradients are not
bit-identical between runs (the dual-stream schedule must not change results).  `python tools/determinism_check.py [B]`.
History: with ds_bpermute-based wave reductions about one LayerNorm-backward row per step came back with a wrong row sum
when the second stream kept other kernels resident on the same CUs; the VALU-only reductions of mbx_common.h fixed it."""
import os, sys
from functools import partial
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from motionbert_amd import DSTformer
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
RUNS = int(os.environ.get('RUNS', 4))
torch.manual_seed(0)
m = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **bench.FULL).cuda()
x, gt = bench.make_batch(B, 243, 17, 5, 'cuda')
runs = []
for r in range(RUNS):
    m.zero_grad(set_to_none=True)
    out = m(x)
    (out * gt).sum().backward()
    torch.cuda.synchronize()
    runs.append((out.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
fail = 0
for r in range(1, RUNS):
    bad = [n for n in runs[0][1] if not torch.equal(runs[r][1][n], runs[0][1][n])]
    same_out = torch.equal(runs[r][0], runs[0][0])
    fail += bool(bad) or not same_out
    print(f'run {r}: out equal {same_out}; {len(bad)} of {len(runs[0][1])} grads differ', bad[:4])
sys.exit(1 if fail else 0)
