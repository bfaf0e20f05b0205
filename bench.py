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
  cpu_baseline  the unmodified reference DSTformer on the host cores when a reference checkout is reachable
                (kind "reference"), else the torch fp32 port of the same path (kind "port"); bounded sample, rank 0, N=1
  forward_only  BASELINE config 1 (eval, no_grad) on the same batch
  gate_1e-3_modes  the same train step in the precisions that meet the north-star 1e-3 gate (bf16x3 split-operand, fp32)
  block_b256    the north-star target configuration: one Block at B=256 (forward / forward+backward, % of MFMA peak)
`python bench.py --gpus N` without a launcher spawns its own N ranks; `--block` prints only the Block benchmark.
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
LITE = dict(FULL, dim_feat=256, mlp_ratio=4)     # MotionBERT-Lite (configs/pretrain/MB_lite.yaml:18-24): C = 256, hidden = 1024, head dim 32
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


LAMBDA_SCALE, LAMBDA_VELOCITY = 0.5, 20.0     # configs/pose3d/MB_train_h36m.yaml:36-43 (the other lambdas are 0)


def pose_loss(pred, gt):
    """loss_mpjpe + 0.5 n_mpjpe + 20 loss_velocity in plain torch (lib/model/loss.py:56-62,81-91,133-142 as combined by
    train.py:176-189): the loss of the CPU baseline.  The GPU step uses the fused kernel motionbert_amd.train.pose_loss,
    which tests/test_gpu_train.py pins against the reference's own loss.py."""
    mpjpe = torch.mean(torch.norm(pred - gt, dim=-1))
    scale = torch.mean(torch.sum(gt * pred, dim=3, keepdim=True), dim=2, keepdim=True) / \
        torch.mean(torch.sum(pred ** 2, dim=3, keepdim=True), dim=2, keepdim=True)
    nm = torch.mean(torch.norm(scale * pred - gt, dim=-1))
    vel = torch.mean(torch.norm((pred[:, 1:] - pred[:, :-1]) - (gt[:, 1:] - gt[:, :-1]), dim=-1)) if pred.shape[1] > 1 else 0.0
    return mpjpe + LAMBDA_SCALE * nm + LAMBDA_VELOCITY * vel


# the NT GEMM entries of the C ABI (same kernels, different epilogues): mbx_gemm_nt, and with LayerNorm folding the GELU' epilogue
# that also emits row dots and the dX GEMM whose epilogue is the LayerNorm backward
NT_FAMILY = ('gemm_nt', 'gemm_nt_dgelu_stats', 'gemm_nt_lnbwd', 'gemm_nt_gelu_d', 'gemm_nt_mul')
# ... and the two N-resident row-owner entries (round 5), which took over the N = 512 products with a row-wise epilogue: with them the
# aggregate below is EVERY A . W^T GEMM launch of a step (the figure used to cover the tile kernels only)
NT_ALL = NT_FAMILY + ('rows_lnbwd_t', 'rows_resid_ln')


class TimedOps:
    """Wraps the kernel provider for ONE instrumented step: HIP events around every C-ABI call."""

    def __init__(self, ops):
        self._ops, self.rec = ops, []

    multi_stream = False   # event pairs must not overlap: the instrumented step runs on one stream

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not callable(fn) or name.startswith('_') or name.startswith('can_'):
            return fn

        def wrapped(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            flops = 0.0
            first = lambda t: t[0] if isinstance(t, tuple) else t      # bf16x3: operands are (hi, lo) plane pairs
            if name in NT_FAMILY:
                flops = 2.0 * first(a[0]).shape[0] * first(a[1]).shape[0] * first(a[0]).shape[1]
            elif name == 'gemm_tn':
                flops = 2.0 * first(a[0]).shape[0] * first(a[0]).shape[1] * first(a[1]).shape[1]
            elif name == 'rows_lnbwd_t':        # (dy [M, K], packed W'^T, xhat [M, 512], ...): the row-owner dX GEMM of a folded pair
                flops = 2.0 * a[0].shape[0] * a[0].shape[1] * a[2].shape[1]
            elif name == 'rows_resid_ln':       # (a [M, K], packed W, bias, resid [M, 512], ...): the row-owner residual GEMM + LayerNorm
                flops = 2.0 * a[0].shape[0] * a[0].shape[1] * a[3].shape[1]
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


def _cpu_name():
    try:
        with open('/proc/cpuinfo') as f:
            return next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
    except Exception:
        return 'unknown'


def _bounded_rate(run, cfg, T, Bc, budget_s):
    """clips/s of `run(T, iters)` (seconds per iteration of Bc clips) on a bounded sample: a T=27 probe sizes it; if a
    full-length batch does not fit the budget the probe length is kept and the rate is converted by the FLOP ratio."""
    run(27, 1)                      # warm-up (thread pool, allocator)
    t27 = run(27, 1)
    ratio = model_flops_fwd(cfg, T) / model_flops_fwd(cfg, 27)
    est_full = t27 * ratio
    log(f'cpu_baseline: B={Bc} T=27 iteration {t27:.2f}s, estimated T={T} iteration {est_full:.1f}s')
    if est_full * 2 <= budget_s:
        iters = max(1, min(4, int(budget_s / est_full) - 1))
        run(T, 1)
        dt = run(T, iters)
        return Bc / dt, f'B={Bc} T={T}, {iters} timed iter(s) after 1 warm-up'
    iters = max(1, min(8, int(budget_s / max(t27, 1e-3))))
    dt = run(27, iters)
    return Bc / (dt * ratio), (f'B={Bc} T=27 x {iters} iter(s) (a T={T} iteration would take ~{est_full:.0f}s), '
                               f'rate converted to T={T} clips by the matmul-FLOP ratio {ratio:.2f}')


def cpu_baseline(cfg, T, budget_s=20.0):
    """The CPU number beside the GPU number.  North star / SURVEY 8d: the UNMODIFIED reference `lib/model/DSTformer.py`
    (fp32, same loss, fwd+bwd) on the host cores -- used whenever a reference checkout is reachable
    ($MOTIONBERT_REFERENCE or /root/reference; it is imported read-only, never copied, bytecode writing off) ->
    kind "reference".  The GPU box has no reference checkout: there a plain torch restatement of the reference model
    (oracle/torch_model.py: the same ATen operator mix and autograd backward) is timed instead -> kind "port", with the measured
    port / reference ratio of the build container beside it (`port_over_reference`); `sample` says which and why."""
    cores = usable_cores()
    torch.set_num_threads(cores)
    ref_dir = os.environ.get('MOTIONBERT_REFERENCE', '/root/reference')
    why_port = f'no reference checkout at {ref_dir} on this host'
    if os.path.isfile(os.path.join(ref_dir, 'lib', 'model', 'DSTformer.py')):
        try:
            sys.dont_write_bytecode = True
            from oracle.make_golden import import_reference
            RefDST = import_reference(ref_dir)
            torch.manual_seed(0)
            m = RefDST(norm_layer=partial(nn.LayerNorm, eps=1e-6), **cfg)
            Bc = 2

            def run(Tc, iters):
                x, gt = make_batch(Bc, Tc, cfg['num_joints'], 1, 'cpu')
                t0 = time.time()
                for _ in range(iters):
                    m.zero_grad(set_to_none=True)
                    pose_loss(m(x), gt).backward()
                return (time.time() - t0) / iters
            value, what = _bounded_rate(run, cfg, T, Bc, budget_s)
            return dict(value=round(value, 4), unit='clips/s', cores=cores, kind='reference',
                        sample=f'unmodified reference lib/model/DSTformer.py from {ref_dir} (fp32, train mode, same pose loss) fwd+bwd, '
                               f'{what}, {cores} threads, CPU: {_cpu_name()}')
        except Exception as e:     # fall through to the port, and say why
            why_port = f'importing the reference from {ref_dir} failed: {type(e).__name__}: {e}'
    # The port: oracle/torch_model.py -- the reference's operator mix (addmm / bmm / softmax / layer-norm / gelu, autograd backward) as
    # one plain torch function over the same parameters.  Same batch, same loss, same thread count as the reference leg above.
    from motionbert_amd import DSTformer
    from oracle import torch_model as TM
    torch.manual_seed(0)
    m = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **cfg)
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    Bc = 2

    def run(Tc, iters):
        x, gt = make_batch(Bc, Tc, cfg['num_joints'], 1, 'cpu')
        t0 = time.time()
        for _ in range(iters):
            for p in P.values():
                p.grad = None
            pose_loss(TM.forward(P, x, cfg['depth'], cfg['num_heads'], eps=1e-6), gt).backward()
        return (time.time() - t0) / iters
    value, what = _bounded_rate(run, cfg, T, Bc, budget_s)
    cal = _port_calibration()
    return dict(value=round(value, 4), unit='clips/s', cores=cores, kind='port', port_over_reference=cal,
                sample=f'plain torch fp32 restatement of the reference model (oracle/torch_model.py: same ATen operators, autograd backward) fwd+bwd '
                       f'({why_port}), {what}, {cores} threads = every core this container may use (affinity mask / cgroup quota; the host has '
                       f'{os.cpu_count()} hardware threads), CPU: {_cpu_name()}; port / unmodified reference on the same cores, legs alternated: '
                       f'{cal["mean"] if cal else "n/a"} +- {cal["spread"] if cal else "n/a"} (tools/cpu_calibration.py in the build container, profiles/r04_cpu_calibration.txt)')


def _port_calibration():
    """{'mean', 'spread', 'n'} of port / reference from the committed calibration log (build container: the GPU box has no reference)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r04_cpu_calibration.txt')) as f:
            d = json.load(f)
        return dict(mean=d['port_over_reference_mean'], spread=d['port_over_reference_spread'], n=len(d['runs']), where='build container')
    except Exception:
        return None


PMC_TABLE = next((p for p in (os.path.join(ROOT, 'profiles', f'r0{r}_pmc_bench.txt') for r in (6, 5, 4, 3, 2)) if os.path.exists(p)),
                 os.path.join(ROOT, 'profiles', 'r03_pmc_bench.txt'))
HBM_PEAK_TBS, HBM_ACHIEVABLE_TBS = 8.0, 6.3      # MI355X_MICROARCH.md: spec / measured float4 copy


def pmc_traffic_bytes(B, T, precision):
    """HBM bytes per launch of the dominant kernel family.  Hardware counters cannot be read from inside the process:
    they come from separate `rocprofv3 --pmc` passes of this very command (tools/gpu_session.sh profiles; FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, as the micro-architecture guide prescribes), summarised by tools/pmc_table.py into profiles/.  The number
    is therefore STATIC (taken from the committed table of this round's kernels, named in `traffic_source`), reported
    only for the configuration the table was taken on, otherwise null."""
    if not (os.path.exists(PMC_TABLE) and B == 64 and T == 243 and precision == 'bf16'):
        return None
    tot, n = 0.0, 0
    for line in open(PMC_TABLE):
        if ('gemm_nt' not in line and 'rows_n_lnbwd' not in line and 'rows_n_resid' not in line) or line.startswith('#'):
            continue
        f = line.split()
        try:     # columns from the right: L2hit% write_MB fetchx2 fetch_MB lds_conf% mfma_busy us n
            write_mb, fetch2_mb, calls = float(f[-2]), float(f[-3]), int(f[-8])
        except (ValueError, IndexError):
            continue
        tot += calls * (fetch2_mb + write_mb) * 1048576    # table is in MiB (counter KB = 1024 B)
        n += calls
    return round(tot / n) if n else None


BLOCK_FLOPS_FWD_PER_CLIP = 36.853e9   # one Block = S sub-block (17.470) + T sub-block (19.383) GFLOP, SURVEY.md 8d


def bench_block(dev, B=256, T=243, iters=5):
    """The north-star target configuration: ONE `Block` (spatial attention + MLP, temporal attention + MLP;
    DSTformer.py:239-244) of the full model at B=256, T=243, J=17, C=512 on one GPU, forward only and forward+backward,
    bf16, timed with HIP events on the launch stream.  Target: >= 50 % of 2.5 PFLOP/s  <=>  forward <= 7.55 ms."""
    from motionbert_amd import DSTformer, hip_ops
    from motionbert_amd.engine import Engine, linear_names
    from motionbert_amd.model import make_cfg
    torch.manual_seed(0)
    model = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **FULL).to(dev)
    cfg = make_cfg(model)
    P = {n: p.detach() for n, p in model.named_parameters()}
    ops = hip_ops.get()
    eng = Engine(ops, cfg, P, torch.bfloat16)
    eng.dual = False
    J, C = cfg.J, cfg.C
    M = B * T * J
    eng.dev, eng.B, eng.Tlen, eng.M = dev, B, T, M
    eng.prepare_weights(True)
    g = torch.Generator(device=dev).manual_seed(3)
    h = torch.randn(M, C, device=dev, generator=g)
    dy = torch.randn(M, C, device=dev, generator=g) * 0.01
    dy_t = dy.to(torch.bfloat16)
    pre = 'blocks_st.0'
    eng.grads = {n: torch.empty_like(p) for n, p in P.items() if n.startswith(pre + '.')}

    def fwd(need_grad):
        # the forward-only pass is the engine's no-grad sequencing (raw-operand LayerNorm + fused MLP) when the provider has it
        raw = (not need_grad) and eng.fold and eng.rawln_allowed and ops.can_fuse_mlp(eng.T, cfg)
        if raw != eng.rawln:
            eng.rawln = raw
            eng.prepare_weights(need_grad)
        return eng._block_fwd(h, pre, 'st', need_grad)

    def fwd_bwd():
        _, svs = fwd(True)
        eng.gstream = eng.fold and eng.gstream_allowed      # (what Engine.backward decides)
        eng._block_bwd(dy, dy_t, svs, pre, 'st', None, last_needs_t=False)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    t_f = timeit(lambda: fwd(False))
    t_fb = timeit(fwd_bwd)
    fl = BLOCK_FLOPS_FWD_PER_CLIP * B
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    return dict(workload=f'one Block (stage_st: S-attn + MLP + T-attn + MLP) of the full model, B={B} T={T} J={J} C={C}, bf16, {iters} timed passes',
                fwd_ms=round(t_f, 3), fwd_tflops=round(fl / t_f / 1e9, 1), fwd_mfma_frac=round(fl / t_f / 1e9 / PEAK_BF16_TFLOPS, 4),
                fwd_bwd_ms=round(t_fb, 3), fwd_bwd_tflops=round(3 * fl / t_fb / 1e9, 1),
                fwd_bwd_mfma_frac=round(3 * fl / t_fb / 1e9 / PEAK_BF16_TFLOPS, 4),
                target='fwd <= 7.55 ms (50 % of 2.5 PFLOP/s: 9.434 TFLOP per forward)', peak_hbm_gib=round(mem, 1))


def _bench_augmenter():
    """Augmenter2D with the reference's own parameter files (params/synthetic_noise.pth, params/d2c_params.pkl) as minted into
    tests/golden/augment2d.npz, and the mask ratios of MB_pretrain.yaml:49-50."""
    import numpy as np
    from motionbert_amd.augment import Augmenter2D
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'augment2d.npz'))
    d = z['d2c']
    return Augmenter2D(noise=dict(mean=torch.from_numpy(z['noise_mean']), std=torch.from_numpy(z['noise_std']), weight=torch.from_numpy(z['noise_weight'])),
                       d2c=dict(a=float(d[0]), b=float(d[1]), m=float(d[2]), s=float(d[3])), mask_ratio=0.05, mask_T_ratio=0.1)


def make_pretrain(net, model, opt, B, J, rank, dev):
    """BASELINE config 4 (MB_pretrain.yaml; train.py:155-206,325-330): one macro step = one PoseTrack-like 2D batch [B,30,17,3]
    (has_gt: mask + noise, loss_2d_weighted), one InstaVariety-like 2D batch [B,81,17,3] (mask only), one 3D batch [B,243,17,3]
    (mask + noise, pose losses) -- three optimizer steps, every piece on the device (motionbert_amd.train.PretrainStep)."""
    from motionbert_amd.train import PretrainStep
    step = PretrainStep(net, opt, aug=_bench_augmenter(), rootrel=True, mask=True, noise=True, lambda_scale=LAMBDA_SCALE, lambda_velocity=LAMBDA_VELOCITY)
    x30, _ = make_batch(B, 30, J, 200 + rank, dev)
    x81, _ = make_batch(B, 81, J, 300 + rank, dev)
    x243, gt243 = make_batch(B, 243, J, 400 + rank, dev)

    def macro():
        step(x30, x30, has_3d=False, has_gt=True)
        step(x81, x81, has_3d=False, has_gt=False)
        return step(x243, gt243, has_3d=True, has_gt=True)
    frames = B * (30 + 81 + 243)
    flops = 3.0 * B * (model_flops_fwd(FULL, 30) + model_flops_fwd(FULL, 81) + model_flops_fwd(FULL, 243))
    return macro, frames, flops


def make_action(model, N, J, rank, dev, distributed):
    """BASELINE config 5 (MB_ft_NTU60_xsub.yaml; train_action.py:143-149,172-188): ActionNet on [N=32, M=2, 243, 17, 3] (backbone
    batch 64), head hidden_dim 2048 / 60 classes / dropout 0.5, cross-entropy, AdamW with lr_backbone 1e-4 and lr_head 1e-3."""
    from motionbert_amd.action import ActionNet
    from motionbert_amd.train import ActionStep
    torch.manual_seed(1)
    net = ActionNet(backbone=model, dim_rep=FULL['dim_rep'], num_classes=60, dropout_ratio=0.5, version='class', hidden_dim=2048,
                    num_joints=J).to(dev).train()
    step = ActionStep(net, lr_backbone=1e-4, lr_head=1e-3, weight_decay=0.01, distributed=distributed)
    x, _ = make_batch(N * 2, 243, J, 500 + rank, dev)
    x = x.reshape(N, 2, 243, J, 3)
    labels = torch.randint(0, 60, (N,), generator=torch.Generator().manual_seed(600 + rank)).to(dev)
    flops = 3.0 * 2 * N * model_flops_fwd(FULL, 243) + 3.0 * 2 * N * (J * FULL['dim_rep'] * 2048 + 2048 * 60)
    return (lambda: step(x, labels)), step, flops


def time_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL rendezvous on
    127.0.0.1), exactly what `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` would do."""
    import socket
    import subprocess
    n = args.gpus
    avail = torch.cuda.device_count()
    if avail < n and args.backend == 'nccl':
        raise SystemExit(f'bench.py --gpus {n}: only {avail} GPU(s) visible')
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = p_.wait() or rc
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=243)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'bf16x3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the forward-only / Block B=256 / fp32-class side measurements')
    ap.add_argument('--block', action='store_true', help='only the north-star Block benchmark (B=256), printed as its own JSON line')
    ap.add_argument('--workload', default='pose', choices=['pose', 'pretrain', 'action'],
                    help="pose: BASELINE configs 2-3 (3D pose train step, the headline); pretrain: config 4 (masked / noisy 2D input, 2D + 3D "
                         "batches of T = 30 / 81 / 243 alternating in lock-step across ranks); action: config 5 (ActionNet finetune, two LR groups)")
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' only for wiring tests)")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        spawn_ranks(args)            # does not return
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
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.block:
        print(json.dumps({'metric': 'DSTformer Block fwd at B=256 (north-star target config)', 'block_b256': bench_block(dev)}), flush=True)
        return

    from motionbert_amd import DSTformer, hip_ops, model as M
    torch.manual_seed(0)
    model = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **FULL).to(dev)
    model.precision = args.precision
    net = model
    if world > 1:
        # one replica per GPU; gradient buckets are all-reduced over RCCL while backward is still running
        from motionbert_amd.ddp import DistributedDSTformer
        net = DistributedDSTformer(model)
    # the training step of train.py:174-206 with its own pieces on the device too: fused pose loss +
    # gradient, one-launch AdamW over the flat parameter buffer (lr 2e-4, wd 0.01: MB_train_h36m.yaml:21-23)
    from motionbert_amd.train import FlatAdamW, GraphedTrainStep, pose_loss as fused_pose_loss
    B, T, J = args.batch, args.frames, FULL['num_joints']
    x, gt = make_batch(B, T, J, 100 + rank, dev)
    wl_flops, wl_units, wl_text = None, B, None        # FLOPs and 243-frame-equivalent clips of one timed step on one rank
    if args.workload == 'action':
        step, action_step, wl_flops = make_action(model, B // 2, J, rank, dev, distributed=world > 1)
        opt = action_step.opt_backbone
        wl_text = (f'BASELINE config 5: ActionNet finetune step (MB_ft_NTU60_xsub.yaml): [{B // 2},2,243,17,3] per GPU -> backbone batch {B}, fused '
                   'dropout(0.5) + mean over T and persons in the backbone tail, fc1 8704->2048 + BatchNorm1d + ReLU + fc2 -> 60 classes, cross-entropy, '
                   'AdamW in two flat groups (lr_backbone 1e-4, lr_head 1e-3); fwd + bwd + both updates timed')
    else:
        opt = FlatAdamW(model, lr=2e-4 if args.workload == 'pose' else 5e-4, weight_decay=0.01)
    if args.workload == 'pretrain':
        step, frames, wl_flops = make_pretrain(net, model, opt, B, J, rank, dev)
        wl_units = frames / 243.0
        wl_text = (f'BASELINE config 4: pre-training macro step (MB_pretrain.yaml; train.py:155-206,325-330) = three optimizer steps per GPU in lock-step '
                   f'across ranks: 2D batch [{B},30,17,3] (mask + synthetic noise, loss_2d_weighted), 2D batch [{B},81,17,3] (mask, loss_2d_weighted), 3D batch '
                   f'[{B},243,17,3] (mask + noise, mpjpe + 0.5 n_mpjpe + 20 velocity); augmentation, losses and AdamW as device kernels; value counts '
                   f'243-frame-equivalent clips ({frames} frames / 243 per macro step per GPU)')

    def pose_step(n=net):
        opt.zero_grad(set_to_none=True)
        total, losses = fused_pose_loss(n(x), gt, LAMBDA_SCALE, LAMBDA_VELOCITY)
        total.backward()
        opt.step()
        return losses
    if args.workload == 'pose':
        step = pose_step

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
    if world > 1 and hasattr(net, 'diagnostics'):
        net.diagnostics = {}
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
    clips = wl_units * world * args.steps / dt
    log(f'timed region: {ms:.2f} ms/step, {clips:.1f} clips/s')

    # ---- N > 1: make the run self-diagnosing -- how many ranks really took part, what the gradient exchange
    # moved and how much of it was NOT hidden under backward, and the same step WITHOUT the exchange on this very GPU in this very
    # process (the N = 1 leg the scaling efficiency is quoted against; weak scaling: the per-GPU work is the same)
    multi = None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        diag = getattr(net, 'diagnostics', None) or {}
        if hasattr(net, 'diagnostics'):
            net.diagnostics = None
        waits = [e0.elapsed_time(e1) for e0, e1 in diag.get('wait_events', [])]
        nb = len(diag.get('bucket_bytes', [])) // max(1, len(waits)) if waits else 0
        single_ms = None
        if args.workload == 'pose':
            for _ in range(2):
                pose_step(model)
            sync()
            t1 = time.perf_counter()
            for _ in range(min(args.steps, 5)):
                pose_step(model)
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t1) / min(args.steps, 5) * 1e3], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            single_ms = float(t.item())
        multi = dict(world_size_seen=int(round(float(ones.item()))), backend=str(dist.get_backend()),
                     allreduce_exposed_ms=round(sum(waits) / len(waits), 3) if waits else None,
                     bucket_bytes=diag.get('bucket_bytes', [])[:nb], buckets_per_step=nb,
                     gradient_bytes_per_step=int(sum(diag.get('bucket_bytes', [])[:nb])),
                     single_gpu_ms_per_step_same_process=round(single_ms, 3) if single_ms else None,
                     scaling_efficiency=round(single_ms / ms, 4) if single_ms else None,
                     note='exposed = time the compute stream waits for the outstanding bucket all-reduces at the end of backward (HIP events '
                          'around the wait, rank 0, mean over the timed steps); single-GPU leg = the same step on the bare replica, no '
                          'gradient exchange, max over ranks; efficiency = single-GPU ms / N-GPU ms (weak scaling)')
        if rank == 0:
            log(f'multi-GPU diagnostics: {multi}')

    # ---- BASELINE config 1 beside the headline: the same model and batch forward-only (eval, no_grad), rank 0 at N=1
    fwd_only = None
    if rank == 0 and world == 1 and not args.no_extras and args.workload == 'pose':
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

    # ---- the mode that meets the north-star 1e-3 gate, with a throughput number beside the headline:
    # the same training step in the fp32-class precisions (same batch, same loss, same optimizer), a short timed sample
    gate_modes = None
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'bf16' and args.workload == 'pose':
        gate_modes = {}
        for prec in ('bf16x3', 'fp32'):
            if prec not in M._DTYPES:
                continue
            model.precision = prec
            try:
                step(); step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                pdt = (time.perf_counter() - t1) / 5
                gate_modes[prec] = dict(value=round(B / pdt, 1), unit='clips/s', ms_per_step=round(pdt * 1e3, 2),
                                        model_tflops=round(3.0 * model_flops_fwd(FULL, T) * B / pdt / 1e12, 1),
                                        sample='same train step (fwd + bwd + AdamW), 5 timed steps after 2 warm-ups',
                                        parity='end-to-end output within 1e-3 of the reference (tests/test_gpu_model.py, fp32-class modes)')
                log(f'{prec}: {pdt * 1e3:.1f} ms/step, {B / pdt:.1f} clips/s')
            except Exception as e:   # e.g. out of memory on a smaller device: report, do not fail the headline
                gate_modes[prec] = dict(error=f'{type(e).__name__}: {e}'[:300])
            torch.cuda.empty_cache()
        model.precision = args.precision
    block = None
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'bf16' and args.workload == 'pose':
        torch.cuda.empty_cache()
        try:
            block = bench_block(dev)
            log(f'Block B=256: fwd {block["fwd_ms"]} ms ({block["fwd_mfma_frac"]:.1%} of MFMA peak), fwd+bwd {block["fwd_bwd_ms"]} ms')
        except Exception as e:
            block = dict(error=f'{type(e).__name__}: {e}'[:300])
        torch.cuda.empty_cache()

    # ---- BASELINE config 3 (MB_train_h36m.yaml:10: batch 32): the step issued eagerly (~850 launches from Python) against the
    # same step replayed from one hipGraph (forward + loss + backward + AdamW captured once)
    cfg3 = None
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'bf16' and args.workload == 'pose':
        try:
            x3, gt3 = make_batch(32, T, J, 7, dev)

            def eager3():
                opt.zero_grad(set_to_none=True)
                total, losses = fused_pose_loss(model(x3), gt3, LAMBDA_SCALE, LAMBDA_VELOCITY)
                total.backward()
                opt.step()

            def timeit(fn, n=5):
                # best of two back-to-back blocks of n steps after three warm-ups: right after empty_cache() the caching allocator
                # can still be growing its pools (hipMalloc is synchronous) -- one run in round 2 showed 774 ms / step for the
                # first block against 70 ms steady state
                fn(); fn(); fn()
                best = float('inf')
                for _ in range(2):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t1) / n * 1e3)
                return best
            e_ms = timeit(eager3)
            gstep = GraphedTrainStep(model, opt, x3, gt3, LAMBDA_SCALE, LAMBDA_VELOCITY)
            g_ms = timeit(lambda: gstep(x3, gt3))
            cfg3 = dict(workload='config 3: full model, B=32 T=243, fwd + fused pose loss + bwd + AdamW, best of two blocks of 5 timed steps after 3 warm-ups',
                        eager_ms=round(e_ms, 2), eager_clips_per_s=round(32e3 / e_ms, 1), graphed_ms=round(g_ms, 2),
                        graphed_clips_per_s=round(32e3 / g_ms, 1))
            log(f'config 3 (B=32): eager {e_ms:.1f} ms, hipGraph replay {g_ms:.1f} ms')
            del gstep
        except Exception as e:
            cfg3 = dict(error=f'{type(e).__name__}: {e}'[:300])
        opt.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()

    # ---- the north-star batch on the FULL model: 256 clips x 243 frames per GPU do not fit with every activation saved
    # (78 GiB at 64 clips -> ~312 GiB); the low-memory mode (model.recompute) rebuilds LayerNorm outputs and MLP post-activations
    full256 = None
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'bf16' and args.workload == 'pose':
        try:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            x4, gt4 = make_batch(256, T, J, 9, dev)
            model.recompute = True

            def step256():
                opt.zero_grad(set_to_none=True)
                total, _l = fused_pose_loss(model(x4), gt4, LAMBDA_SCALE, LAMBDA_VELOCITY)
                total.backward()
                opt.step()
            step256()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                step256()
            torch.cuda.synchronize()
            d256 = (time.perf_counter() - t1) / 5
            full256 = dict(workload='full model train step (fwd + loss + bwd + AdamW), 256 clips x 243 frames on ONE GPU, recompute mode, 5 timed steps after 1 warm-up',
                           ms_per_step=round(d256 * 1e3, 1), clips_per_s=round(256 / d256, 1), model_tflops=round(3.0 * model_flops_fwd(FULL, T) * 256 / d256 / 1e12, 1),
                           peak_hbm_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
            log(f'full model B=256 (recompute): {d256 * 1e3:.0f} ms/step, {256 / d256:.1f} clips/s, peak HBM {full256["peak_hbm_gib"]} GiB')
            del x4, gt4
        except Exception as e:
            full256 = dict(error=f'{type(e).__name__}: {e}'[:300])
        model.recompute = False
        opt.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()

    # ---- MotionBERT-Lite (the reference's second published architecture: MB_lite.yaml) at the headline batch: the same training step
    # and the no-grad forward on the C = 256 / head-dim-32 kernels (fixtures-tested, never timed before round 4)
    lite = None
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'bf16' and args.workload == 'pose':
        try:
            torch.cuda.empty_cache()
            torch.manual_seed(1)
            mlite = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **LITE).to(dev)
            mlite.precision = 'bf16'
            olite = FlatAdamW(mlite, lr=2e-4, weight_decay=0.01)

            def lite_step():
                olite.zero_grad(set_to_none=True)
                total, _l = fused_pose_loss(mlite(x), gt, LAMBDA_SCALE, LAMBDA_VELOCITY)
                total.backward()
                olite.step()
            for _ in range(2):
                lite_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                lite_step()
            torch.cuda.synchronize()
            ldt = (time.perf_counter() - t1) / 5
            mlite.eval()
            with torch.no_grad():
                for _ in range(2):
                    mlite(x)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    mlite(x)
                torch.cuda.synchronize()
            lfd = (time.perf_counter() - t1) / 5
            lf = model_flops_fwd(LITE, T) * B
            lite = dict(workload=f'MotionBERT-Lite (dim_feat 256, mlp_ratio 4; 16.0 M parameters), {B} clips x {T} frames, bf16: train step (fwd + loss + bwd + AdamW) and eval + no_grad forward, 5 timed passes each after 2 warm-ups',
                        train_ms_per_step=round(ldt * 1e3, 2), train_clips_per_s=round(B / ldt, 1), train_model_tflops=round(3.0 * lf / ldt / 1e12, 1),
                        fwd_ms=round(lfd * 1e3, 2), fwd_clips_per_s=round(B / lfd, 1), fwd_tflops=round(lf / lfd / 1e12, 1))
            log(f'Lite B={B}: train {ldt * 1e3:.1f} ms/step ({B / ldt:.0f} clips/s), forward only {lfd * 1e3:.2f} ms ({B / lfd:.0f} clips/s)')
            del mlite, olite
        except Exception as e:
            lite = dict(error=f'{type(e).__name__}: {e}'[:300])
        torch.cuda.empty_cache()

    # ---- BASELINE configs 4 and 5 at N = 1 beside the headline; `--workload pretrain|action` times them as
    # the main step (and under --gpus N), these are short samples of the same step functions
    cfg4 = cfg5 = None
    if rank == 0 and world == 1 and not args.no_extras and args.precision == 'bf16' and args.workload == 'pose':
        try:
            macro, frames, fl4 = make_pretrain(model, model, opt, B, J, rank, dev)
            d4 = time_steps(macro, 6, 2)
            cfg4 = dict(workload=f'config 4 (MB_pretrain.yaml): macro step = 2D batch [{B},30,17,3] (mask + noise, loss_2d_weighted) + 2D batch [{B},81,17,3] '
                                 f'(mask, loss_2d_weighted) + 3D batch [{B},243,17,3] (mask + noise, pose losses): three fwd + bwd + AdamW steps, augment2D / losses '
                                 'as device kernels; 6 timed macro steps after 2 warm-ups',
                        ms_per_macro_step=round(d4 * 1e3, 2), frames_per_s=round(frames / d4, 1), clips243_equiv_per_s=round(frames / 243.0 / d4, 1),
                        model_tflops=round(fl4 / d4 / 1e12, 1))
            log(f'config 4 (pretrain macro step): {d4 * 1e3:.1f} ms, {frames / 243.0 / d4:.1f} 243-frame-equivalent clips/s')
            del macro
        except Exception as e:
            cfg4 = dict(error=f'{type(e).__name__}: {e}'[:300])
        opt.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()
        try:
            torch.manual_seed(0)
            m5 = DSTformer(norm_layer=partial(nn.LayerNorm, eps=1e-6), **FULL).to(dev)
            m5.precision = args.precision
            fn5, st5, fl5 = make_action(m5, 32, J, rank, dev, distributed=False)
            d5 = time_steps(fn5, 8, 2)
            cfg5 = dict(workload='config 5 (MB_ft_NTU60_xsub.yaml): ActionNet step on [32,2,243,17,3] (backbone batch 64), dropout 0.5 + means fused in the '
                                 'backbone tail, head 8704->2048->60, cross-entropy, two flat AdamW groups (lr 1e-4 / 1e-3); 8 timed steps after 2 warm-ups',
                        ms_per_step=round(d5 * 1e3, 2), samples_per_s=round(32 / d5, 1), clips_per_s=round(64 / d5, 1), model_tflops=round(fl5 / d5 / 1e12, 1))
            log(f'config 5 (ActionNet step): {d5 * 1e3:.1f} ms, {64 / d5:.1f} clips/s')
            del fn5, st5, m5
        except Exception as e:
            cfg5 = dict(error=f'{type(e).__name__}: {e}'[:300])
        torch.cuda.empty_cache()

    # ---- one extra instrumented step (untimed): per-kernel HIP-event durations -> roofline of the dominant kernel
    roof, breakdown = None, None
    if rank == 0:
        timed = TimedOps(hip_ops.get())
        for _ in range(2):      # the first pass refills the allocator's pools after empty_cache() (a hipMalloc inside an event pair
            timed.rec.clear()   # once put 25 ms on the LayerNorm-backward line); the second pass is the one reported
            opt.zero_grad(set_to_none=True)
            total, _ = fused_pose_loss(M.run(timed, model, x), gt, LAMBDA_SCALE, LAMBDA_VELOCITY)
            total.backward()
            torch.cuda.synchronize()
        agg = timed.summary()
        tot = sum(d['ms'] for d in agg.values())
        breakdown = {k: dict(calls=d['calls'], ms=round(d['ms'], 3), share=round(d['ms'] / tot, 4)) for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
        dom = max(agg, key=lambda k: agg[k]['ms'])
        d = dict(calls=0, ms=0.0, flops=0.0)
        for k in NT_ALL:
            for f in d:
                d[f] += agg.get(k, {}).get(f, 0)
        peak = PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
        traffic = pmc_traffic_bytes(B, T, args.precision)
        fpl, avg_s = d['flops'] / d['calls'], d['ms'] / d['calls'] * 1e-3
        # which roof bounds the family: its arithmetic intensity (algorithmic FLOPs over the HBM bytes the counters saw) against
        # the ridge peak / achievable-HBM (2500 TFLOP/s / 6.3 TB/s ~ 400 FLOP/B); without a counter table the label stays "mfma"
        intensity = fpl / traffic if traffic else None
        bound = 'hbm' if intensity is not None and intensity < peak / HBM_ACHIEVABLE_TBS else 'mfma'
        hbm_tbs = traffic / avg_s / 1e12 if traffic else None
        roof = dict(bound=bound, kernel='every A . W^T GEMM launch of a step (`launches`): mbx_gemm_nt / mbx_gemm_nt_gelu_d / mbx_gemm_nt_mul -> gemm_nt_pp256_kernel (store, GELU + GELU\' saved, multiply epilogues) / gemm_nt_pipe_kernel (residual epilogue); mbx_rows_lnbwd_t / mbx_rows_resid_ln -> rows_n_lnbwd_kernel / rows_n_resid_ln_kernel (N-resident row owners: LayerNorm backward / residual + next LayerNorm as epilogue); bf16 MFMA', achieved=round(ach, 1), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4),
                    traffic=traffic, traffic_unit='HBM bytes per launch (launch-weighted mean over the gemm_nt and rows_n kernels)',
                    traffic_source=('STATIC: ' + os.path.relpath(PMC_TABLE, ROOT) + ' (separate rocprofv3 --pmc passes of this command on this round\'s kernels: '
                                    'FETCH_SIZE x2 + WRITE_SIZE); not measured by this run') if os.path.exists(PMC_TABLE) else None, launches=d['calls'], avg_launch_ms=round(d['ms'] / d['calls'], 4),
                    flops_per_launch=fpl, flop_per_hbm_byte=round(intensity, 1) if intensity else None,
                    hbm_tb_s=round(hbm_tbs, 3) if hbm_tbs else None, hbm_frac=round(hbm_tbs / HBM_PEAK_TBS, 4) if hbm_tbs else None,
                    hbm_frac_of_achievable=round(hbm_tbs / HBM_ACHIEVABLE_TBS, 4) if hbm_tbs else None,
                    attainable_tflops=round(min(peak, intensity * HBM_ACHIEVABLE_TBS), 1) if intensity else None, dominant_by_time=dom)
        # the same figure per C-ABI GEMM entry: since round 3 the family above also carries what used to be stand-alone HBM passes
        # (the LayerNorm backward is the epilogue of mbx_gemm_nt_lnbwd), so its FLOP rate dropped while the step got shorter
        # the ceiling the part actually offers under its power cap: nothing but MFMAs in a loop, random operands, ~0.3 s (VERDICT r4
        # item 7); `frac` above stays quoted against the datasheet peak, `frac_of_sustained` against this measurement of the same run
        try:
            pr = hip_ops.get().mfma_probe(0.3) if args.precision == 'bf16' else None
            if pr is not None:
                roof.update(sustained_mfma_tflops=round(pr['tflops'], 1), sustained_clock_ghz=round(pr['clock_ghz'], 3),
                            frac_of_sustained=round(ach / pr['tflops'], 4),
                            sustained_source=f"mbx_mfma_probe: {pr['iters']} x 16 v_mfma_f32_32x32x16_bf16 per wave, one wave per SIMD on every CU, pseudo-random "
                                             f"operands, one launch of {pr['ms']:.0f} ms timed with HIP events in this run; clock = shader cycles / real time inside the kernel")
        except Exception as e:
            roof['sustained_mfma_tflops'] = None
            roof['sustained_error'] = f'{type(e).__name__}: {e}'[:200]
        roof['by_entry'] = {k: dict(launches=agg[k]['calls'], ms=round(agg[k]['ms'], 3), tflops=round(agg[k]['flops'] / (agg[k]['ms'] * 1e-3) / 1e12, 1),
                                    frac=round(agg[k]['flops'] / (agg[k]['ms'] * 1e-3) / 1e12 / peak, 4))
                            for k in NT_FAMILY + ('gemm_tn', 'rows_lnbwd_t', 'rows_resid_ln') if k in agg and agg[k]['ms'] > 0}
    flops_step = wl_flops if wl_flops is not None else 3.0 * model_flops_fwd(FULL, T) * B
    out = {
        'metric': 'clips/sec [B,243,17,3] DSTformer fwd+bwd', 'value': round(clips, 2), 'unit': 'clips/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': wl_text if wl_text else f'MotionBERT full DSTformer (dim_feat 512, depth 5, 8 heads, mlp_ratio 2) train step: fwd + bwd + AdamW, '
                               f'{B} clips/GPU x T={T} x J={J}, random-init weights, pose loss (mpjpe + 0.5 n_mpjpe + 20 velocity, fused kernel), one-launch flat AdamW',
                   'global_batch': B * world, 'frames': T, 'parallelism': f'dp{world}'},
        'model_tflops': round(flops_step * world / (ms * 1e-3) / 1e12, 1),
        'model_mfma_frac': round(flops_step / (ms * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS), 4),
        'multi_gpu': multi, 'roofline': roof, 'kernel_breakdown_ms': breakdown, 'forward_only': fwd_only, 'gate_1e-3_modes': gate_modes, 'block_b256': block, 'config3_b32': cfg3, 'full_model_b256': full256, 'lite_b64': lite, 'config4_pretrain': cfg4, 'config5_action': cfg5,
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
