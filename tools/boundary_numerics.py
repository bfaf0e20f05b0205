"""Round 5 (experiment with a kill criterion): the gradient of the residual stream in bf16 ACROSS Block boundaries (levels >= 1): the first
sub-layer of a Block takes the row-owner LayerNorm backward like the inner ones (bf16 out, no second summand), and the fusion backward of
the level below adds the two Blocks' input gradients from a bf16 pair.  CPU run of the product's own sequencing (engine.py) with the torch
restatement of the kernel set (oracle/torch_ops.MockOps, bf16) on the reference-minted fixtures, four rounding realisations, switch off /
on (MBX_BLOCK_GRAD_T), against the EXISTING gates of tests/test_gpu_model.py::test_baseline_shape_fixture_fwd_bwd.
    python tools/boundary_numerics.py [fixture]      -> profiles/r05_boundary_numerics.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import engine, model as M              # noqa: E402
from oracle.torch_ops import MockOps                       # noqa: E402
from tests.helpers import build_model, load_golden, rel_l2, trained_like   # noqa: E402
from tests.test_gpu_model import _fixture_grad_errors      # noqa: E402

torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else 'full_1x243'
z, cfg = load_golden(name)
names = [str(n) for n in z['names']]
ac_per = dict(zip(names, (float(a) for a in z['autocast_grad_per'])))
ac_out, ac_glob = float(z['autocast_out']), float(z['autocast_grad_global'])

for pseed in (0, 1, 2, 3):
    for tag, sw in (('fp32 across Block boundaries', '0'), ('bf16 across Block boundaries', '1')):
        os.environ['MBX_BLOCK_GRAD_T'] = sw
        engine.reload_switches()      # (the switches are read once at import)
        model = build_model(cfg, seed=0)
        if int(z['trained_seed']) >= 0:
            trained_like(model, int(z['trained_seed']))
        if pseed:
            g = torch.Generator().manual_seed(100 + pseed)
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1 + 1e-6 * torch.randn(p.shape, generator=g))
        model.precision = 'bf16'
        ops = MockOps()
        x = torch.from_numpy(z['x']).requires_grad_(True)
        t0 = time.time()
        out = M.run(ops, model, x)
        (out * torch.from_numpy(z['cot'])).sum().backward()
        e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
        e_dx = rel_l2(x.grad.numpy(), z['dx'])
        bad = {n: round(per[n], 4) for n in names if per[n] > max(3 * ac_per[n], 0.08)}
        ok = e_all < min(2 * ac_glob, max(0.08, ac_glob)) and not bad
        print(f'seed {pseed} {tag}: out {rel_l2(out.detach().numpy(), z["out"]):.4f} dx {e_dx:.4f} grad_global {e_all:.4f} (reference under autocast '
              f'{ac_glob:.4f}) worst {worst} {e_worst:.4f}  pair launches {ops.calls.count("fuse_bwd_pair")}  gates {"PASS" if ok else "FAIL " + str(bad)}  '
              f'[{time.time() - t0:.0f}s]', flush=True)
