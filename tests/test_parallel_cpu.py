"""CPU: the data-parallel host logic (gradient buckets over gloo, world_size 2; FlatAdam == torch Adam)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from stp3_amd.parallel import FlatAdam, GradientBuckets


def _toy():
    torch.manual_seed(3)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1),
                         nn.Flatten(), nn.Linear(4 * 6 * 6, 5))


def test_flat_adam_matches_torch_adam_and_params_alias_buckets():
    a, b = _toy(), _toy()
    b[0].weight.data = b[0].weight.data.contiguous(memory_format=torch.channels_last)
    buckets = GradientBuckets(b, bucket_bytes=2048)
    assert len(buckets.buckets) > 1
    for p in b.parameters():                       # parameters and gradients live inside the flat buffers
        assert any(p.data_ptr() >= f.data_ptr() and p.data_ptr() < f.data_ptr() + f.numel() * 4
                   for f in buckets.flat_params)
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=1e-3)
    opt_b = FlatAdam(buckets, lr=1e-2, weight_decay=1e-3)
    g = torch.Generator().manual_seed(0)
    for _ in range(4):
        x = torch.randn(6, 3, 6, 6, generator=g)
        opt_a.zero_grad()
        a(x).square().mean().backward()
        opt_a.step()
        buckets.zero_grad()
        b(x).square().mean().backward()
        buckets.finish()
        opt_b.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pb, pa, rtol=1e-4, atol=2e-5)   # Adam amplifies 1-ulp gradient differences where g ~ 0


def test_clip_grad_norm_matches_torch():
    a, b = _toy(), _toy()
    buckets = GradientBuckets(b, bucket_bytes=4096)
    x = torch.randn(4, 3, 6, 6)
    a(x).square().sum().backward()
    buckets.zero_grad()
    b(x).square().sum().backward()
    na = torch.nn.utils.clip_grad_norm_(a.parameters(), 0.5)
    nb = buckets.clip_grad_norm_(0.5)
    torch.testing.assert_close(nb, na, rtol=1e-5, atol=1e-6)
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pb.grad, pa.grad, rtol=1e-5, atol=1e-7)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # different initial weights: broadcast must fix that
    model = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 3))
    buckets = GradientBuckets(model, bucket_bytes=256)
    opt = FlatAdam(buckets, lr=1e-2)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(8, 6, generator=g)
    target = torch.randn(8, 3, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)              # each rank sees half of the global batch
    for _ in range(3):
        buckets.zero_grad()
        loss = (model(data[shard]) - target[shard]).square().mean()
        loss.backward()
        buckets.finish()
        opt.step()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_concatenated_batch():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    torch.testing.assert_close(out[0], out[1], rtol=0, atol=0)         # replicas stay identical
    # single process, full batch, same initial weights as rank 0
    torch.manual_seed(100)
    model = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 3))
    buckets = GradientBuckets(model, bucket_bytes=256)
    opt = FlatAdam(buckets, lr=1e-2)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(8, 6, generator=g)
    target = torch.randn(8, 3, generator=g)
    for _ in range(3):
        buckets.zero_grad()
        (model(data) - target).square().mean().backward()
        buckets.finish()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(out[0], ref, rtol=1e-5, atol=1e-6)


def _bn_worker(rank, world, port, out):
    from stp3_amd.layers import fused
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(5)
    bn = nn.BatchNorm2d(6)
    x = torch.randn(4, 6, 5, 7)
    gy = torch.randn(4, 6, 5, 7)
    sl = slice(2 * rank, 2 * rank + 2)
    xa = x[sl].clone().requires_grad_(True)
    y = fused.bn_act(bn, xa, fused.ACT_RELU)
    y.backward(gy[sl])
    out[rank] = (y.detach(), xa.grad, bn.running_var.clone())
    dist.destroy_process_group()


def test_cross_replica_batchnorm_statistics_on_cpu():
    """bn_act with 2 ranks x half the batch == 1 process x whole batch (the reference's sync_batchnorm)."""
    from stp3_amd.layers import fused
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bn_worker, args=(2, port, out), nprocs=2, join=True)
    torch.manual_seed(5)
    bn = nn.BatchNorm2d(6)
    x = torch.randn(4, 6, 5, 7)
    gy = torch.randn(4, 6, 5, 7)
    xa = x.clone().requires_grad_(True)
    y = fused.bn_act(bn, xa, fused.ACT_RELU)
    y.backward(gy)
    torch.testing.assert_close(torch.cat([out[0][0], out[1][0]]), y.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(torch.cat([out[0][1], out[1][1]]), xa.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out[0][2], bn.running_var, rtol=1e-5, atol=1e-6)
