"""The MLP sub-layer of the no-grad path, fused against unfused, at the bench shapes (through the C ABI, HIP events on the launch
stream): ms per launch, TFLOP/s of the two GEMMs, algorithmic HBM bytes.  `python tools/mlp_bench.py [clips ...]`."""
import sys

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from motionbert_amd import hip_ops                      # noqa: E402
from motionbert_amd.engine import EPI_GELU, EPI_RESID   # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ops = hip_ops.get()
    dev = 'cuda'
    for clips in [int(a) for a in sys.argv[1:]] or [64, 256]:
        for C, hidden in ((512, 1024), (256, 1024)):
            M = clips * 243 * 17
            g = torch.Generator(device=dev).manual_seed(0)
            x = torch.randn(M, C, device=dev, generator=g)
            a = x.to(BF)
            xh = torch.empty(M, C, device=dev, dtype=BF)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            w1 = (torch.randn(hidden, C, device=dev, generator=g) * 0.05).to(BF)
            w2 = (torch.randn(C, hidden, device=dev, generator=g) * 0.05).to(BF)
            b1, b2 = torch.randn(hidden, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
            rsum = w1.float().sum(1)
            packed = ops.mlp_pack_weights(w1, w2)
            y, yt = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev, dtype=BF)
            gbuf = torch.empty(M, hidden, device=dev, dtype=BF)
            flops = 4.0 * M * C * hidden
            t_ln = timeit(lambda: ops.layernorm_fwd(x, None, None, 1e-6, xh, mean, rstd))
            t_fc1 = timeit(lambda: ops.gemm_nt(xh, w1, b1, EPI_GELU, out_t=None, out2_t=gbuf))
            t_fc2 = timeit(lambda: ops.gemm_nt(gbuf, w2, b2, EPI_RESID, out_f=y, resid=x))
            t_f0 = timeit(lambda: ops.mlp_fused_fwd(xh, 0, packed, b1, b2, None, x, y, None, 1e-6, None, None))
            t_f1 = timeit(lambda: ops.mlp_fused_fwd(a, 1, packed, b1, b2, rsum, x, y, yt, 1e-6, mean, rstd))
            t_pack = timeit(lambda: ops.mlp_pack_weights(w1, w2))
            gb_f = (2 + 4 + 4 + 2) * M * C / 1e9
            print(f'clips={clips} C={C} hidden={hidden} M={M}: unfused ln {t_ln:.3f} + fc1 {t_fc1:.3f} + fc2 {t_fc2:.3f} = {t_ln + t_fc1 + t_fc2:.3f} ms '
                  f'({flops / (t_fc1 + t_fc2) / 1e9:.0f} TF/s GEMMs) | fused(xhat in, y out) {t_f0:.3f} ms = {flops / t_f0 / 1e9:.0f} TF/s | '
                  f'fused(raw in, y + bf16 y + stats out) {t_f1:.3f} ms = {flops / t_f1 / 1e9:.0f} TF/s, {gb_f / t_f1:.2f} TB/s algorithmic | pack {t_pack * 1e3:.1f} us',
                  flush=True)


if __name__ == '__main__':
    main()
