"""GEMM micro-benchmark on the GPU box: the model's GEMM shapes through the C ABI, TF/s per shape
(HIP events on the launch stream, random data), with a correctness spot check against torch.

    python tools/gemm_bench.py [--batch 64] [--iters 20]       (MBX_GEMM_V1=1 selects the simple kernels)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionbert_amd import hip_ops
from motionbert_amd.engine import EPI_GELU, EPI_RESID, EPI_STORE, EPI_DGELU


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--check', type=int, default=1)
    ap.add_argument('--only', default='', help='comma list of shape names')
    args = ap.parse_args()
    ops = hip_ops.get()
    M = args.batch * 243 * 17
    dev, bf = 'cuda', torch.bfloat16
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    res = {}
    # forward / dX shapes (N, K, epilogue)
    nt = [('qkv', 1536, 512, EPI_STORE), ('proj', 512, 512, EPI_RESID), ('fc1', 1024, 512, EPI_GELU), ('fc2', 512, 1024, EPI_RESID),
          ('dX_qkv', 512, 1536, EPI_STORE), ('dX_fc2', 1024, 512, EPI_DGELU), ('dX_fc1', 512, 1024, EPI_STORE),
          # diagnostics (not model shapes): fc1's shape with a plain store / with GELU but without the pre-activation output
          ('st1024', 1024, 512, EPI_STORE), ('fc1_noU', 1024, 512, EPI_GELU)]
    only = set(filter(None, args.only.split(',')))
    for name, N, K, epi in nt:
        if only and name not in only:
            continue
        a, w, bias = rnd(M, K).to(bf), (rnd(N, K) * 0.05).to(bf), rnd(N)
        out_t, out2, out_f = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev)
        resid, aux = rnd(M, N), rnd(M, N).to(bf)
        kw = dict(out_t=out_t)
        if epi == EPI_GELU:
            kw = dict(out_t=out_t if name != 'fc1_noU' else None, out2_t=out2)
        elif epi == EPI_RESID:
            kw = dict(out_f=out_f, resid=resid)
        elif epi == EPI_DGELU:
            kw = dict(out_t=out_t, aux_t=aux)
        fn = lambda: ops.gemm_nt(a, w, bias if epi != EPI_DGELU else None, epi, **kw)
        ms = timeit(fn, args.iters)
        tf = 2.0 * M * N * K / ms / 1e9
        err = None
        if args.check:
            rows = torch.randint(0, M, (512,), generator=g).to(dev)
            ref = a[rows].float() @ w.float().t()
            if epi != EPI_DGELU:
                ref = ref + bias
            if name == 'fc1_noU':
                got, ref = out2[rows].float(), torch.nn.functional.gelu(ref)
            elif epi == EPI_STORE or epi == EPI_GELU:
                got = out_t[rows].float()
            elif epi == EPI_RESID:
                got, ref = out_f[rows], ref + resid[rows]
            else:
                u = aux[rows].float()
                gp = 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * 3.141592653589793) ** 0.5
                got, ref = out_t[rows].float(), ref * gp
            err = float((got - ref).norm() / ref.norm())
        res['nt.' + name] = dict(ms=round(ms, 4), tflops=round(tf, 1), err=err)
        print(f'nt {name:8s} M={M} N={N} K={K}: {ms:8.4f} ms  {tf:7.1f} TF/s  err={err}', flush=True)
        del a, w, out_t, out2, out_f, resid, aux
    # dX GEMMs with the LayerNorm-backward epilogue (round 3): d(qkv) -> dx, d(fc1) -> dx
    for name, N, K in [('lnb_qkv', 512, 1536), ('lnb_fc1', 512, 1024)]:
        if only and name not in only:
            continue
        a, w = rnd(M, K).to(bf), (rnd(N, K) * 0.05).to(bf)
        xhat, dres, rowc = rnd(M, N).to(bf), rnd(M, N), torch.rand(M, 4, generator=g).to(dev)
        dx, dx_t = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev, dtype=bf)
        ms = timeit(lambda: ops.gemm_nt_lnbwd(a, w, xhat, rowc, dres, None, dx, dx_t), args.iters)
        tf = 2.0 * M * N * K / ms / 1e9
        err = None
        if args.check:
            rows = torch.randint(0, M, (512,), generator=g).to(dev)
            acc = a[rows].float() @ w.float().t()
            rc = rowc[rows]
            ref = dres[rows] + rc[:, 0:1] * acc - rc[:, 1:2] - xhat[rows].float() * rc[:, 2:3]
            err = float((dx[rows] - ref).norm() / ref.norm())
        res['nt.' + name] = dict(ms=round(ms, 4), tflops=round(tf, 1), err=err)
        print(f'nt {name:8s} M={M} N={N} K={K}: {ms:8.4f} ms  {tf:7.1f} TF/s  err={err}', flush=True)
        del a, w, xhat, dres, rowc, dx, dx_t
    tn = [('dW_qkv', 1536, 512), ('dW_proj', 512, 512), ('dW_fc1', 1024, 512), ('dW_fc2', 512, 1024)]
    for name, N, K in tn:
        if only and name not in only:
            continue
        dy, a = rnd(M, N).to(bf), rnd(M, K).to(bf)
        dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        ms = timeit(lambda: ops.gemm_tn(dy, a, dw, db), args.iters)
        tf = 2.0 * M * N * K / ms / 1e9
        err = None
        if args.check:
            Ms = min(M, 200000)
            dw2 = torch.empty(N, K, device=dev)
            ops.gemm_tn(dy[:Ms], a[:Ms], dw2, db)
            ref = dy[:Ms].float().t() @ a[:Ms].float()
            err = float((dw2 - ref).norm() / ref.norm())
            errb = float((db - dy[:Ms].float().sum(0)).norm() / dy[:Ms].float().sum(0).norm())
            err = max(err, errb)
        res['tn.' + name] = dict(ms=round(ms, 4), tflops=round(tf, 1), err=err)
        print(f'tn {name:8s} M={M} N={N} K={K}: {ms:8.4f} ms  {tf:7.1f} TF/s  err={err}', flush=True)
        del dy, a
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    tag = 'v1' if os.environ.get('MBX_GEMM_V1') == '1' else 'pipe'
    with open(os.path.join(out, f'gemm_bench_{tag}.json'), 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
