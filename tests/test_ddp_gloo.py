"""The N>1 path on CPU: world_size 2 over gloo.  Two processes hold identical replicas and half of the
batch each; the bucketed, overlapped gradient all-reduce of motionbert_amd.ddp must reproduce the
gradients of one process running the whole batch (mean-reduced loss, equal shards).  The kernels are the
torch restatement (test infrastructure): this test is about the distributed wiring only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import build_model, load_golden, make_input


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from motionbert_amd.ddp import DistributedDSTformer
        from oracle.torch_ops import MockOps
        z, cfg = load_golden('tiny_trained')
        model = build_model(cfg, seed=100 + rank)          # different init per rank: broadcast must fix it
        if rank == 0:
            model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')})
        model.precision = 'fp32'
        ddp = DistributedDSTformer(model, ops=MockOps())
        x = make_input(4, 9, 17, 21)
        tgt = torch.randn(4, 9, 17, 3, generator=torch.Generator().manual_seed(22))
        lo, hi = rank * 2, rank * 2 + 2
        out = ddp(x[lo:hi])
        loss = ((out - tgt[lo:hi]) ** 2).mean()
        loss.backward()
        q.put((rank, {n: p.grad.numpy().copy() for n, p in model.named_parameters()},
               {n: p.detach().numpy().copy() for n, p in model.named_parameters()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradients_match_single_process():
    from motionbert_amd import model as M
    from oracle.torch_ops import MockOps
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, g, w = q.get(timeout=240)
        res[r] = (g, w)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch
    z, cfg = load_golden('tiny_trained')
    model = build_model(cfg)
    model.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')})
    model.precision = 'fp32'
    x = make_input(4, 9, 17, 21)
    tgt = torch.randn(4, 9, 17, 3, generator=torch.Generator().manual_seed(22))
    loss = ((M.run(MockOps(), model, x) - tgt) ** 2).mean()
    loss.backward()
    for n, p in model.named_parameters():
        ref = p.grad.numpy()
        for r in range(world):
            assert np.array_equal(res[r][1][n], p.detach().numpy()), f'rank {r} did not receive rank 0 weights for {n}'
            assert np.allclose(res[r][0][n], ref, rtol=2e-4, atol=1e-7), (n, r)
        assert np.array_equal(res[0][0][n], res[1][0][n]), f'ranks disagree on {n}'


def test_bucket_layout_follows_backward_order():
    from motionbert_amd.engine import grad_bucket
    assert grad_bucket('head.weight', 5) == 0 and grad_bucket('norm.bias', 5) == 0
    assert grad_bucket('blocks_ts.4.mlp_t.fc1.weight', 5) == 1 and grad_bucket('ts_attn.4.bias', 5) == 1
    assert grad_bucket('blocks_st.0.attn_s.qkv.bias', 5) == 5
    assert grad_bucket('temp_embed', 5) == 6 and grad_bucket('joints_embed.weight', 5) == 6


def _head_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import torch.nn as nn
        from motionbert_amd.ddp import DistributedDSTformer
        from oracle.torch_ops import MockOps
        z, cfg = load_golden('tiny_trained')
        torch.manual_seed(200 + rank)                       # different init per rank: the broadcast must fix backbone AND head
        backbone = build_model(cfg)
        head = nn.Sequential(nn.Linear(17 * cfg['dim_rep'], 32), nn.BatchNorm1d(32), nn.ReLU(), nn.Linear(32, 5))
        head[1].running_mean.fill_(float(rank))             # a buffer that must arrive from rank 0
        head.eval()                                          # BN on running statistics: per-rank batches then add up exactly
        backbone.precision = 'fp32'
        ddp = DistributedDSTformer(backbone, ops=MockOps(), extra=head)
        x = make_input(4, 9, 17, 31)
        labels = torch.tensor([1, 4, 0, 2])
        lo, hi = rank * 2, rank * 2 + 2
        if rank == 0:
            rep = ddp.get_representation(x[lo:hi])                              # [2, 9, 17, R]  (model_action.py:68)
        else:
            # attach(): a model that owns the backbone and calls IT (ActionNet.backbone) -- the sync rides on the module
            from motionbert_amd import model as Mm
            ddp.attach()
            assert backbone._grad_sync.pending is ddp._pending
            rep = Mm.run(MockOps(), backbone, x[lo:hi], True, backbone._grad_sync)
            ddp.detach()
            assert not hasattr(backbone, '_grad_sync')
        logits = head(rep.mean(1).reshape(2, -1))                                # mean over T, joints flattened (:20-24)
        loss = torch.nn.functional.cross_entropy(logits, labels[lo:hi])
        loss.backward()
        q.put((rank, {n: p.grad.numpy().copy() for n, p in head.named_parameters()},
               {n: (None if p.grad is None else p.grad.numpy().copy()) for n, p in backbone.named_parameters()},
               {n: p.detach().numpy().copy() for n, p in head.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_actionnet_head_gradients_are_synchronised():
    """BASELINE config 5 (ActionNet finetune): the head's parameters live outside the backbone.  With `extra=head` the wrapper
    must broadcast them (and the BatchNorm buffers) and average their gradients; the backbone's `head.*` gets no gradient on
    the representation path; everything equals one process on the whole batch."""
    import torch.nn as nn
    from motionbert_amd import model as M
    from oracle.torch_ops import MockOps
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_head_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, gh, gb, sd = q.get(timeout=240)
        res[r] = (gh, gb, sd)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch, rank 0's initial weights (seed 200)
    z, cfg = load_golden('tiny_trained')
    torch.manual_seed(200)
    backbone = build_model(cfg)
    head = nn.Sequential(nn.Linear(17 * cfg['dim_rep'], 32), nn.BatchNorm1d(32), nn.ReLU(), nn.Linear(32, 5))
    head.eval()
    backbone.precision = 'fp32'
    x = make_input(4, 9, 17, 31)
    rep = M.run(MockOps(), backbone, x, return_rep=True)
    loss = torch.nn.functional.cross_entropy(head(rep.mean(1).reshape(4, -1)), torch.tensor([1, 4, 0, 2]))
    loss.backward()
    for r in range(world):
        assert np.array_equal(res[r][2]['1.running_mean'], np.zeros(32, np.float32)), 'BatchNorm buffer was not broadcast from rank 0'
        for n, p in head.named_parameters():
            assert np.array_equal(res[r][2][n], p.detach().numpy()), f'rank {r}: head parameter {n} was not broadcast'
            assert np.allclose(res[r][0][n], p.grad.numpy(), rtol=2e-4, atol=1e-7), (n, r)
        for n, p in backbone.named_parameters():
            if n.startswith('head.'):
                assert res[r][1][n] is None and p.grad is None
            else:
                assert np.allclose(res[r][1][n], p.grad.numpy(), rtol=2e-4, atol=1e-7), (n, r)
    for n in res[0][0]:
        assert np.array_equal(res[0][0][n], res[1][0][n]), f'ranks disagree on head gradient {n}'


def _lockstep_worker(rank, world, port, q):
    """One pre-training epoch over three loaders of DIFFERENT lengths (n clips not divisible by world x batch) on world ranks."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from motionbert_amd.data import shard_indices
        from motionbert_amd.ddp import DistributedDSTformer
        from motionbert_amd.train import pretrain_epoch_plan
        from oracle.torch_ops import MockOps
        z, cfg = load_golden('tiny_trained')
        model = build_model(cfg, seed=300 + rank)
        model.precision = 'fp32'
        ddp = DistributedDSTformer(model, ops=MockOps())
        sizes, frames, batch = dict(posetrack=7, instav=10, **{'3d': 13}), dict(posetrack=5, instav=9, **{'3d': 9}), 2
        data = {k: make_input(n, frames[k], 17, 40 + i) for i, (k, n) in enumerate(sizes.items())}
        counts = {k: -(-len(shard_indices(n, True, 0, 0, rank, world)) // batch) for k, n in sizes.items()}
        steps = []
        for loader, has_3d, has_gt, nb in pretrain_epoch_plan(counts['posetrack'], counts['instav'], counts['3d'], epoch=30):
            idx = shard_indices(sizes[loader], True, 0, 0, rank, world)
            for b in range(nb):
                xb = data[loader][torch.from_numpy(idx[b * batch:(b + 1) * batch].copy())]
                out = ddp(xb)
                model.zero_grad(set_to_none=True)
                ((out - xb) ** 2).mean().backward()                        # any loss: the exchange is what is tested
                with torch.no_grad():
                    for p in model.parameters():
                        p.sub_(0.05 * p.grad)
                steps.append((loader, tuple(xb.shape)))
        q.put((rank, steps, {n: p.detach().numpy().copy() for n, p in model.named_parameters()}, sorted(np.concatenate(
            [shard_indices(sizes['3d'], True, 0, 0, rank, world)]).tolist())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_four_ranks_stay_in_lock_step_over_unequal_loaders():
    """BASELINE config 4 on N ranks (SURVEY 8e gotcha 4; train.py:325-330): the three loaders of a pre-training epoch have different
    lengths and none divides by world x batch.  Equal wrap-around shards (`shard_indices`) + one epoch plan (`pretrain_epoch_plan`)
    make every rank issue the same steps in the same order -- the run neither dead-locks in a collective nor lets the replicas
    drift: after the epoch the parameters of the four ranks are bit-identical, and the shards cover every clip."""
    world, port = 4, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_lockstep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, steps, params, idx3d = q.get(timeout=500)
        res[r] = (steps, params, idx3d)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kinds = [[s[0] for s in res[r][0]] for r in range(world)]
    assert all(k == kinds[0] for k in kinds), 'ranks walked different step sequences'
    assert kinds[0] == ['posetrack'] * 1 + ['instav'] * 2 + ['3d'] * 2      # ceil(ceil(n / 4) / 2) steps per loader
    for n in res[0][1]:
        for r in range(1, world):
            assert np.array_equal(res[0][1][n], res[r][1][n]), f'rank {r} drifted on {n}'
    assert set(sum((res[r][2] for r in range(world)), [])) == set(range(13)), 'the shards do not cover the 3D loader'
