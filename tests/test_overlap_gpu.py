"""GPU: the bucketed gradient all-reduce runs WHILE backward continues (reference recipe: Lightning DDP,
/root/reference/train.py:43-56 -- gradient buckets reduced as they fill).

Two ranks on ONE GPU.  RCCL refuses that ("Duplicate GPU detected : rank 1 and rank 0 both on CUDA device",
scripts/probe_nccl_one_gpu.py, run on the MI355X box), so the process group is gloo (CUDA tensors travel through
the host); the bucket logic under test -- post-accumulate hooks, reverse-forward bucket order, asynchronous launch
from the hook of a bucket's last gradient -- is backend-independent.  Evidence, from event timestamps and work
handles of rank 0:
  * every bucket but the last is launched from a hook, i.e. before ``backward()`` has returned;
  * the first bucket's gradient is complete (HIP event on the compute stream at its launch) after less than a
    third of backward's GPU time;
  * its all-reduce has COMPLETED while the GPU is still busy with backward (``wait()`` returns, the end-of-backward
    event has not fired yet)."""
import os
import socket

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from stp3_amd.parallel import GradientBuckets
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    layers = []
    for _ in range(24):
        layers += [nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.ReLU()]
    model = nn.Sequential(*layers).cuda()
    buckets = GradientBuckets(model, bucket_bytes=64 << 10)           # one 147 KB weight per bucket
    n = len(buckets.buckets)
    x = torch.randn(8, 64, 192, 192, device='cuda')
    launches = {}
    inner = buckets._launch
    state = {'in_backward': False}

    def launch(i):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        launches[i] = (ev, state['in_backward'])
        inner(i)

    buckets._launch = launch
    for step in range(2):                                             # step 0 warms up (MIOpen find, allocator)
        launches.clear()
        buckets.zero_grad()
        loss = model(x).square().mean()
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        state['in_backward'] = True
        loss.backward()
        state['in_backward'] = False
        end.record()
        first_work = buckets._works[0]
        first_work.wait()
        gpu_still_busy = not end.query()
        buckets.finish()
        torch.cuda.synchronize()
    total = start.elapsed_time(end)
    first = start.elapsed_time(launches[0][0])
    out[rank] = dict(n=n, from_hook=sum(1 for _, h in launches.values() if h), first_ms=first, total_ms=total,
                     done_while_busy=bool(gpu_still_busy),
                     grads_equal=None)
    # both ranks end with the same averaged gradient
    flat = torch.cat([f for f, _ in buckets.buckets])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    out[rank] = dict(out[rank], grads_equal=bool(torch.equal(gathered[0], gathered[1])))
    dist.destroy_process_group()


def test_bucket_all_reduce_overlaps_backward():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = dict(out[0])
    print(r)
    assert r['n'] == 24 and r['from_hook'] >= r['n'] - 1, r            # launched during backward, not after it
    assert r['first_ms'] < r['total_ms'] / 3, r                          # first bucket ready early in backward
    assert r['done_while_busy'], r                                      # and reduced while the GPU still computes
    assert r['grads_equal'] and out[1]['grads_equal']
