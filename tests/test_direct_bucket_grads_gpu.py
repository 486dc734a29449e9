"""GPU: weight gradients written straight into their bucket slice (ops.DIRECT_BUCKET_GRADS) and their split-K sums deferred to
one launch at the end of backward (ops.DEFER_WGRAD_REDUCE) leave the flat gradient buckets BIT-EQUAL to the plain route
(fresh gradient tensors, gathered by GradientBuckets).  The reference has no counterpart -- PyTorch-Lightning's DDP keeps
per-parameter ``.grad`` tensors (/root/reference/train.py:43-56) -- the buckets are what this repo's optimizer reads.

Covered: a weight used by one layer, a weight shared by three applications in one graph (the GRU cells of the prediction
stage: only the FIRST contribution may claim the slice, autograd adds the others), a convolution whose output channels are
padded to a multiple of 8 (its gradient is a slice of a padded tensor and must take the gather route), biases (never direct),
and the post-accumulate hooks of world size 2 (two gloo ranks on one GPU: the deferred route must stay off there, a bucket is
all-reduced the moment its last gradient lands)."""
import os
import socket

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(16, 32, 3, padding=1, bias=True)
        self.shared = nn.Conv2d(32, 32, 3, padding=1, bias=False)
        self.odd = nn.Conv2d(32, 35, 1, bias=False)                   # Cout padded to 40 inside the operator
        self.tail = nn.Conv2d(32, 64, 1, bias=False)

    def forward(self, x):
        from stp3_amd.layers import fused
        h = fused.conv2d(x, self.a.weight, self.a.bias, 1, 1, 1)
        for _ in range(3):
            h = torch.relu(fused.conv2d(h, self.shared.weight, None, 1, 1, 1))
        return fused.conv2d(h, self.odd.weight, None, 1, 0, 1).float().square().mean() + \
            fused.conv2d(h, self.tail.weight, None, 1, 0, 1).float().square().mean()


def _run(direct, defer, hooks_seen=None):
    from stp3_amd import ops
    from stp3_amd.parallel import GradientBuckets
    ops.DIRECT_BUCKET_GRADS, ops.DEFER_WGRAD_REDUCE = direct, defer
    torch.manual_seed(3)
    model = _Net().cuda().to(memory_format=torch.channels_last)       # the layout of dw: what the product's models hold
    buckets = GradientBuckets(model, bucket_bytes=32 << 10)
    x = torch.randn(4, 16, 24, 40, generator=torch.Generator().manual_seed(9)).cuda().to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    out = []
    for _ in range(2):                                                # the second pass meets claims of the first
        buckets.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = model(x)
        loss.backward()
        if buckets.world == 1:
            # after backward every gradient is either the alias of its slice or a tensor the gather copies
            for views, (_, params) in zip(buckets.grad_views, buckets.buckets):
                for v, p in zip(views, params):
                    assert p.grad is not None
                    alias = p.grad.data_ptr() == v.data_ptr()
                    if p is model.shared.weight:
                        continue        # the engine's sum of three contributions: in place in the first (the alias) or fresh
                    assert alias == (direct and (p is model.a.weight or p is model.tail.weight)), (p.shape, alias)
            assert (ops.pending_wgrad_reductions() > 0) == (direct and defer)
        buckets.finish()
        assert ops.pending_wgrad_reductions() == 0
        out.append(torch.cat([f.clone() for f, _ in buckets.buckets]))
    return out


def test_direct_and_deferred_weight_gradients_bit_equal():
    from stp3_amd import ops
    keep = ops.DIRECT_BUCKET_GRADS, ops.DEFER_WGRAD_REDUCE
    try:
        plain = _run(False, False)
        direct = _run(True, False)
        deferred = _run(True, True)
    finally:
        ops.DIRECT_BUCKET_GRADS, ops.DEFER_WGRAD_REDUCE = keep
    assert float(plain[0].abs().max()) > 0
    for a, b, c in zip(plain, direct, deferred):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(plain[0], plain[1])                            # same input, same weights: the passes agree too


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from stp3_amd import ops
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    plain = _run(False, False)
    direct = _run(True, True)                                          # DEFER is requested and must be ignored with 2 ranks
    out[rank] = bool(all(torch.equal(a, b) for a, b in zip(plain, direct))) and ops.pending_wgrad_reductions() == 0
    dist.destroy_process_group()


def test_direct_weight_gradients_under_bucket_hooks_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    out = ctx.Manager().dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] and out[1], dict(out)
