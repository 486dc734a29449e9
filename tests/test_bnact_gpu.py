"""GPU parity of the fused BatchNorm + activation (+ residual) HIP operators (through the C ABI) against the
plain-torch float32 statement of the same op (``layers.fused.bn_act_reference``).

Tolerances: float32 I/O  rtol 1e-4 / atol 1e-5 (different summation order of the batch statistics);
bfloat16 I/O: outputs are compared after rounding the reference to bf16, rtol 2e-2 / atol 2e-2
(one bf16 ulp at |y| ~ 4); parameter gradients accumulate bf16-rounded terms -> rtol 3e-2.
"""
import os
import socket

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _mk(c, n, h, w, dtype, sliced, seed):
    g = torch.Generator().manual_seed(seed)
    cp = (c + 7) // 8 * 8 if sliced else c
    x = (torch.randn(n, cp, h, w, generator=g) * 1.5 + 0.3).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    res = torch.randn(n, c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    sbias = torch.randn(n, c, generator=g).cuda()
    oscale = (torch.rand(n, generator=g) > 0.3).float().cuda() / 0.7
    gy = torch.randn(n, c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(c, momentum=0.05, eps=1e-3).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return x, res, sbias, oscale, gy, bn


CASES = [
    # c, n, h, w, act, res_mode, sbias, oscale, sliced
    (64, 3, 20, 24, 'relu', 'none', False, False, False),
    (48, 4, 28, 30, 'swish', 'none', False, False, False),
    (24, 4, 16, 20, 'none', 'after', False, True, False),      # MBConv projection + drop-connect + skip
    (128, 2, 25, 25, 'relu', 'before', False, False, False),   # ResNet BasicBlock tail
    (64, 3, 40, 40, 'relu', 'after', True, False, False),      # TemporalBlock aggregation: pyramid bias + skip
    (35, 3, 40, 40, 'relu', 'none', False, False, True),       # odd channel count, channel-sliced conv output
    (960, 5, 14, 30, 'swish', 'none', False, False, False),    # widest trunk layer
    (8, 2, 200, 200, 'relu', 'none', True, False, False),      # few channels, many rows
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('case', CASES)
def test_bn_act_matches_reference(case, train, dtype):
    from stp3_amd.layers import fused
    c, n, h, w, act, res_mode, use_sb, use_os, sliced = case
    act = {'none': fused.ACT_NONE, 'relu': fused.ACT_RELU, 'swish': fused.ACT_SWISH}[act]
    res_mode = {'none': fused.RES_NONE, 'before': fused.RES_BEFORE_ACT, 'after': fused.RES_AFTER_ACT}[res_mode]
    x, res, sbias, oscale, gy, bn = _mk(c, n, h, w, dtype, sliced, seed=c + n)
    ref_bn = nn.BatchNorm2d(c, momentum=0.05, eps=1e-3).cuda()
    ref_bn.load_state_dict(bn.state_dict())
    bn.train(train)
    ref_bn.train(train)

    def leaves(*ts):
        return [None if t is None else t.detach().clone().requires_grad_(True) for t in ts]

    xa, ra, sa = leaves(x, res if res_mode else None, sbias if use_sb else None)
    xb, rb, sb = leaves(x.float(), res.float() if res_mode else None, sbias if use_sb else None)
    xin_a = xa[:, :c] if sliced else xa
    xin_b = xb[:, :c] if sliced else xb
    ya = fused.bn_act(bn, xin_a, act, res=ra, res_mode=res_mode, sbias=sa, oscale=oscale if use_os else None)
    yb = fused.bn_act_reference(ref_bn, xin_b, act, res=rb, res_mode=res_mode, sbias=sb,
                                oscale=oscale if use_os else None)
    assert ya.dtype == dtype and ya.shape == yb.shape
    ya.backward(gy)
    yb.backward(gy.float())
    if dtype == torch.float32:
        tol = dict(rtol=1e-4, atol=1e-5)
        ptol = dict(rtol=1e-4, atol=1e-4)
    else:
        tol = dict(rtol=2e-2, atol=2e-2)
        ptol = dict(rtol=3e-2, atol=3e-2 * (n * h * w) ** 0.5)
    torch.testing.assert_close(ya.float(), yb, **tol)
    torch.testing.assert_close(xa.grad.float()[:, :c], xb.grad[:, :c], **tol)
    if sliced:
        assert (xa.grad[:, c:] == 0).all()
    torch.testing.assert_close(bn.weight.grad, ref_bn.weight.grad, **ptol)
    torch.testing.assert_close(bn.bias.grad, ref_bn.bias.grad, **ptol)
    if ra is not None:
        torch.testing.assert_close(ra.grad.float(), rb.grad, **tol)
    if sa is not None:
        torch.testing.assert_close(sa.grad, sb.grad, **ptol)
    if train:
        torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bn.running_var, ref_bn.running_var, rtol=1e-3 if dtype == torch.float32 else 2e-2,
                                   atol=1e-4)
        from stp3_amd import ops
        ops.flush_batch_counters()        # the counter is kept on the host and applied at flush / state_dict() time
        assert int(bn.num_batches_tracked) == 1


def test_bn_act_is_deterministic_and_rejects_cpu():
    from stp3_amd import ops
    from stp3_amd.layers import fused
    x, res, sbias, oscale, gy, bn = _mk(64, 4, 50, 50, torch.bfloat16, False, seed=9)
    a = fused.bn_act(bn, x, fused.ACT_SWISH)
    b = fused.bn_act(bn, x, fused.ACT_SWISH)
    assert torch.equal(a, b)
    with pytest.raises(Exception):
        ops.bn_act(x.cpu(), bn.weight, bn.bias, bn.running_mean, bn.running_var, True, 0.1, 1e-5)


def _sync_worker(rank, world, port, out):
    import torch.distributed as dist
    from stp3_amd.layers import fused
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)     # gloo moves CUDA tensors through the host
    torch.cuda.set_device(0)
    x, res, sbias, oscale, gy, bn = _mk(48, 4, 12, 10, torch.float32, False, seed=5)
    sl = slice(2 * rank, 2 * rank + 2)
    xa = x[sl].detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = fused.bn_act(bn, xa, fused.ACT_RELU)
    y.backward(gy[sl])
    out[rank] = (y.detach().cpu(), xa.grad.cpu(), bn.weight.grad.cpu(), bn.running_var.cpu())
    dist.destroy_process_group()


def test_cross_replica_statistics_two_ranks_one_gpu():
    """Two ranks (both on cuda:0, gloo) with half the batch each == one rank with the whole batch."""
    import torch.multiprocessing as mp
    from stp3_amd.layers import fused
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_sync_worker, args=(2, port, out), nprocs=2, join=True)
    x, res, sbias, oscale, gy, bn = _mk(48, 4, 12, 10, torch.float32, False, seed=5)
    xa = x.detach().clone().requires_grad_(True)
    y = fused.bn_act(bn, xa, fused.ACT_RELU)
    y.backward(gy)
    torch.testing.assert_close(torch.cat([out[0][0], out[1][0]]), y.detach().cpu(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(torch.cat([out[0][1], out[1][1]]), xa.grad.cpu(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out[0][2] + out[1][2], bn.weight.grad.cpu(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out[0][3], bn.running_var.cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_backward_reads_a_channel_slice_of_the_gradient_in_place(dtype):
    """The gradient of a ``torch.cat`` arrives as a channel slice of a wider channels-last tensor: the backward kernels
    read it with its own row stride (no dense copy) and give the same bits as for a dense gradient."""
    from stp3_amd.layers import fused
    x, res, sbias, oscale, gy, bn = _mk(64, 3, 20, 24, dtype, False, seed=3)
    wide = torch.randn(3, 192, 20, 24, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
    wide[:, 64:128] = gy
    out = []
    for g in (gy, wide[:, 64:128]):
        xa = x.detach().clone().requires_grad_(True)
        bn.zero_grad()
        fused.bn_act(bn, xa, fused.ACT_SWISH).backward(g)
        out.append((xa.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
    assert not wide[:, 64:128].is_contiguous(memory_format=torch.channels_last)
    for a, b in zip(*out):
        assert torch.equal(a, b)
