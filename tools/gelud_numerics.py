"""VERDICT r4 item 5 (experiment with a kill criterion): fc1's forward epilogue saves bf16(gelu'(u)), taken from the fp32 accumulator,
INSTEAD of bf16(u); the backward epilogue is one multiply (mbx_gemm_nt_gelu_d / mbx_gemm_nt_mul).  CPU emulation with the torch
restatement of the kernel set (oracle/torch_ops.MockOps, bf16) on the reference-minted fixtures, four rounding realisations, against
the EXISTING gates of tests/test_gpu_model.py::test_baseline_shape_fixture_fwd_bwd.
    python tools/gelud_numerics.py [fixture]      -> profiles/r05_gelud_numerics.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motionbert_amd import model as M                      # noqa: E402
from oracle.torch_ops import MockOps                       # noqa: E402
from tests.helpers import build_model, load_golden, rel_l2, trained_like   # noqa: E402
from tests.test_gpu_model import _fixture_grad_errors      # noqa: E402

torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else 'full_1x243'
z, cfg = load_golden(name)
names = [str(n) for n in z['names']]
ac_per = dict(zip(names, (float(a) for a in z['autocast_grad_per'])))
ac_out, ac_glob = float(z['autocast_out']), float(z['autocast_grad_global'])
BF = torch.bfloat16


for pseed in (0, 1, 2, 3):
    for tag in ('u saved', 'd saved'):
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
        ops.fuse_gelu_d = tag == 'd saved'
        x = torch.from_numpy(z['x']).requires_grad_(True)
        t0 = time.time()
        out = M.run(ops, model, x)
        (out * torch.from_numpy(z['cot'])).sum().backward()
        e_all, e_worst, worst, e_norm, per = _fixture_grad_errors(model, z)
        e_dx = rel_l2(x.grad.numpy(), z['dx'])
        bad = {n: round(per[n], 4) for n in names if per[n] > max(3 * ac_per[n], 0.08)}
        ok = e_all < min(2 * ac_glob, max(0.08, ac_glob)) and not bad
        print(f'seed {pseed} {tag}: out {rel_l2(out.detach().numpy(), z["out"]):.4f} dx {e_dx:.4f} grad_global {e_all:.4f} (reference under autocast '
              f'{ac_glob:.4f}) worst {worst} {e_worst:.4f}  gates {"PASS" if ok else "FAIL " + str(bad)}  [{time.time() - t0:.0f}s]', flush=True)
