"""bench.py -- clips/sec of the DSTformer hot path (fwd + bwd + AdamW) on N MI355X of one node.

    python bench.py                               # N=1, defaults finish in a couple of minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic clips resident in HBM:
forward + backward of the full MotionBERT backbone (configs/pretrain/MB_pretrain.yaml:18-24 ==
configs/pose3d/MB_train_h36m.yaml: dim_feat 512, depth 5, 8 heads, mlp_ratio 2, T=243, J=17) at
64 clips per GPU (MB_pretrain.yaml:10), a pose loss, and the AdamW update -- nothing is skipped
inside the timed region.  One process per GPU; gradients are all-reduced over RCCL (weak scaling:
the per-GPU batch is fixed).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant kernel (the bf16 MFMA GEMM gemm_nt): algorithmic FLOPs per launch /
                average launch duration measured here with HIP events on the launch stream during one
                extra, untimed, instrumented step; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)
  cpu_baseline  the torch fp32 port of the same path (oracle/torch_ops.py) timed on the host cores of
                this node on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from functools import partial

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.nn as nn

FULL = dict(dim_in=3, dim_out=3, dim_feat=512, dim_rep=512, depth=5, num_heads=8, mlp_ratio=2, num_joints=17, maxlen=243)
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3


def model_flops_fwd(cfg, T):
    """Matmul FLOPs of one clip forward (closed form of SURVEY.md 8d / BASELINE.md section 2)."""
    C, R, J, H = cfg['dim_feat'], cfg['dim_rep'], cfg['num_joints'], cfg['num_heads']
    hid = int(C * cfg['mlp_ratio'])
    N = T * J
    lin = 2 * N * C * (3 * C + C + 2 * hid)
    sp, tm = 4 * T * J * J * C, 4 * J * T * T * C
    return cfg['depth'] * (2 * (2 * lin + sp + tm)) + cfg['depth'] * 2 * N * 2 * C * 2 + 2 * N * 3 * C + 2 * N * C * R + 2 * N * R * 3


def make_batch(B, T, J, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.cat([torch.rand(B, T, J, 2, generator=g) * 2 - 1, torch.rand(B, T, J, 1, generator=g)], -1)
    gt = torch.randn(B, T, J, 3, generator=g) * 0.3
    gt = gt - gt[:, :, 0:1]
    return x.to(device), gt.to(device)


def pose_loss(pred, gt):
    """loss_mpjpe + 20 * loss_velocity, restated (lib/model/loss.py:56-66,133-142; MB_train_h36m.yaml:38-39)."""
    mpjpe = torch.mean(torch.norm(pred - gt, dim=-1))
    vel = torch.mean(torch.norm((pred[:, 1:] - pred[:, :-1]) - (gt[:, 1:] - gt[:, :-1]), dim=-1)) if pred.shape[1] > 1 else 0.0
    return mpjpe + 20.0 * vel


class TimedOps:
    """Wraps the kernel provider for ONE instrumented step: HIP events around every C-ABI call."""

    def __init__(self, ops):
        self._ops, self.rec = ops, []

    multi_stream = False   # event pairs must not overlap: the instrumented step runs on one stream

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not callable(fn) or name.startswith('_'):
            return fn

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            flops = 0.0
            if name == 'gemm_nt':
                flops = 2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1]
            elif name == 'gemm_tn':
                flops = 2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[1]
            self.rec.append((name, flops, e0, e1))
            return r
        return wrapped

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, e0, e1 in self.rec:
            d = agg.setdefault(name, dict(calls=0, ms=0.0, flops=0.0))
            d['calls'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['flops'] += flops
        return agg


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()
            if q != 'max':
                n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = max(1, min(n, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(cfg, T, budget_s=20.0):
    """Torch fp32 port of the hot path on the host cores (oracle/torch_ops.py), fwd+bwd of ONE clip.
    Bounded: a T=27 probe sizes the sample; if a full-length clip does not fit the budget the probe
    length is kept and the rate is converted to full-length clips by the FLOP ratio (stated in `sample`)."""
    from motionbert_amd import DSTformer, model as M
    from oracle.torch_ops import MockOps
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **cfg)
    m.precision = 'fp32'

    def run(Tc, iters):
        x, gt = make_batch(1, Tc, cfg['num_joints'], 1, 'cpu')
        t0 = time.time()
        for _ in range(iters):
            m.zero_grad(set_to_none=True)
            pose_loss(M.run(MockOps(), m, x), gt).backward()
        return (time.time() - t0) / iters
    run(27, 1)                      # warm-up (thread pool, allocator)
    t27 = run(27, 1)
    ratio = model_flops_fwd(cfg, T) / model_flops_fwd(cfg, 27)
    est_full = t27 * ratio
    log(f'cpu_baseline: {cores} usable cores, T=27 clip {t27:.2f}s, estimated T={T} clip {est_full:.1f}s')
    if est_full * 2 <= budget_s:
        iters = max(1, min(4, int(budget_s / est_full) - 1))
        run(T, 1)
        dt = run(T, iters)
        value, what = 1.0 / dt, f'B=1 T={T}, {iters} timed iter(s) after 1 warm-up'
    else:
        iters = max(1, min(8, int(budget_s / max(t27, 1e-3))))
        dt = run(27, iters)
        value, what = 1.0 / (dt * ratio), (f'B=1 T=27 x {iters} iter(s) (a T={T} clip would take ~{est_full:.0f}s), '
                                           f'rate converted to T={T} clips by the matmul-FLOP ratio {ratio:.2f}')
    cpu = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
    except Exception:
        pass
    return dict(value=round(value, 4), unit='clips/s', cores=cores, kind='port',
                sample=f'torch fp32 port (oracle/torch_ops.py) of the full model fwd+bwd, {what}, {cores} threads, CPU: {cpu}')


def pmc_traffic_bytes(B, T, precision):
    """HBM bytes per launch of the dominant kernel family from the committed PMC summary (separate rocprofv3 --pmc
    passes of this very workload, corrected as the micro-architecture guide prescribes: FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE).  Counters cannot be read from inside the process, so the number is the profile's, not live: it is
    reported only for the configuration the profile was taken on, otherwise null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_bench_v6.txt')
    if not (os.path.exists(path) and B == 64 and T == 243 and precision == 'bf16'):
        return None
    tot, n = 0.0, 0
    for line in open(path):
        if 'gemm_nt_pipe' not in line:
            continue
        f = line.split()
        try:     # columns from the right: L2hit% write_MB fetchx2 fetch_MB lds_conf% mfma_busy us n
            write_mb, fetch2_mb, calls = float(f[-2]), float(f[-3]), int(f[-8])
        except (ValueError, IndexError):
            continue
        tot += calls * (fetch2_mb + write_mb) * 1048576    # table is in MiB (counter KB = 1024 B)
        n += calls
    return round(tot / n) if n else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=243)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' only for wiring tests)")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a ROCm device (the hot path has no CPU implementation)'
    local = local % torch.cuda.device_count()   # (wiring tests may oversubscribe one GPU with the gloo backend)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'

    from motionbert_amd import DSTformer, hip_ops, model as M
    torch.manual_seed(0)
    model = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **FULL).to(dev)
    model.precision = args.precision
    net = model
    if world > 1:
        # one replica per GPU; gradient buckets are all-reduced over RCCL while backward is still running
        from motionbert_amd.ddp import DistributedDSTformer
        net = DistributedDSTformer(model)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0.01, fused=True)
    B, T, J = args.batch, args.frames, FULL['num_joints']
    x, gt = make_batch(B, T, J, 100 + rank, dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = pose_loss(net(x), gt)
        loss.backward()
        opt.step()
        return loss

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log(f'rank {rank}/{world}: model on {dev}, B={B} T={T} precision={args.precision}')
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log(f'first step done, HBM in use {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    clips = B * world * args.steps / dt
    log(f'timed region: {ms:.2f} ms/step, {clips:.1f} clips/s')

    # ---- BASELINE config 1 beside the headline: the same model and batch forward-only (eval, no_grad), rank 0 at N=1
    fwd_only = None
    if rank == 0 and world == 1:
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                model(x)
            torch.cuda.synchronize()
        fdt = (time.perf_counter() - t1) / 5
        model.train()
        fwd_only = dict(value=round(B / fdt, 1), unit='clips/s', ms=round(fdt * 1e3, 2),
                        tflops=round(model_flops_fwd(FULL, T) * B / fdt / 1e12, 1), sample='eval + no_grad, 5 timed passes after 2 warm-ups')
        log(f'forward only: {fdt * 1e3:.2f} ms, {B / fdt:.1f} clips/s')

    # ---- one extra instrumented step (untimed): per-kernel HIP-event durations -> roofline of the dominant kernel
    roof, breakdown = None, None
    if rank == 0:
        timed = TimedOps(hip_ops.get())
        opt.zero_grad(set_to_none=True)
        loss = pose_loss(M.run(timed, model, x), gt)
        loss.backward()
        agg = timed.summary()
        tot = sum(d['ms'] for d in agg.values())
        breakdown = {k: dict(calls=d['calls'], ms=round(d['ms'], 3), share=round(d['ms'] / tot, 4)) for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
        dom = max(agg, key=lambda k: agg[k]['ms'])
        d = agg['gemm_nt']
        peak = PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
        roof = dict(bound='mfma', kernel='mbx_gemm_nt -> gemm_nt_pipe256_kernel / gemm_nt_pipe_kernel (bf16 MFMA GEMM, all 162 launches of a step)', achieved=round(ach, 1), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4),
                    traffic=pmc_traffic_bytes(B, T, args.precision), traffic_unit='HBM bytes per launch (launch-weighted mean over the gemm_nt kernels)',
                    traffic_ref='profiles/r01_pmc_bench_v6.txt (rocprofv3 --pmc passes of this command: FETCH_SIZE x2 + WRITE_SIZE, KB)', launches=d['calls'], avg_launch_ms=round(d['ms'] / d['calls'], 4),
                    flops_per_launch=d['flops'] / d['calls'], dominant_by_time=dom)
    flops_step = 3.0 * model_flops_fwd(FULL, T) * B
    out = {
        'metric': 'clips/sec [B,243,17,3] DSTformer fwd+bwd', 'value': round(clips, 2), 'unit': 'clips/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': f'MotionBERT full DSTformer (dim_feat 512, depth 5, 8 heads, mlp_ratio 2) train step: fwd + bwd + AdamW, '
                               f'{B} clips/GPU x T={T} x J={J}, random-init weights, pose loss (mpjpe + 20 velocity)',
                   'global_batch': B * world, 'frames': T, 'parallelism': f'dp{world}'},
        'model_tflops': round(flops_step * world / (ms * 1e-3) / 1e12, 1),
        'model_mfma_frac': round(flops_step / (ms * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS), 4),
        'roofline': roof, 'kernel_breakdown_ms': breakdown, 'forward_only': fwd_only,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(FULL, T)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
