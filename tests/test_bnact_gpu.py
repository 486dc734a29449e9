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


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('mode', ['relu', 'swish_before', 'relu_after_sbias'])
def test_zero_padded_channel_lanes(mode, train, dtype):
    """stp3_bn_dims.cpad: a 35-channel layer in 40-lane rows.  The padding lanes of x / res / dy hold NaN; the real
    channels must match the float32 torch statement and the padding lanes of y / dx (/ dres) must be exactly zero."""
    from stp3_amd.layers import fused
    n, c, cp, h, w = 6, 35, 40, 50, 40
    act, res_mode, use_sb = {'relu': (fused.ACT_RELU, fused.RES_NONE, False),
                             'swish_before': (fused.ACT_SWISH, fused.RES_BEFORE_ACT, False),
                             'relu_after_sbias': (fused.ACT_RELU, fused.RES_AFTER_ACT, True)}[mode]
    x, res, sbias, _, gy, bn = _mk(c, n, h, w, dtype, False, seed=77)
    ref_bn = nn.BatchNorm2d(c, momentum=0.05, eps=1e-3).cuda()
    ref_bn.load_state_dict(bn.state_dict())
    bn.train(train)
    ref_bn.train(train)

    def padded(t):
        full = torch.full((n, cp, h, w), float('nan'), dtype=dtype, device='cuda').contiguous(memory_format=torch.channels_last)
        full[:, :c] = t
        return full
    xk = padded(x).requires_grad_()
    rk = padded(res).requires_grad_() if res_mode != fused.RES_NONE else None
    sbk = sbias.clone().requires_grad_() if use_sb else None
    y = fused.bn_act(bn, xk, act, rk, res_mode, sbk)
    y.backward(padded(gy))
    xr = x.float().requires_grad_()
    rr = res.float().requires_grad_() if rk is not None else None
    sbr = sbias.clone().requires_grad_() if use_sb else None
    yr = fused.bn_act_reference(ref_bn, xr, act, rr, res_mode, sbr)
    yr.backward(gy.float())
    assert y.shape == (n, cp, h, w) and y.is_contiguous(memory_format=torch.channels_last)
    assert bool((y.detach()[:, c:] == 0).all()) and bool((xk.grad[:, c:] == 0).all())
    if res_mode == fused.RES_BEFORE_ACT:
        assert bool((rk.grad[:, c:] == 0).all())
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    ptol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(y.detach()[:, :c].float(), yr.detach().to(dtype).float(), **tol)
    torch.testing.assert_close(xk.grad[:, :c].float(), xr.grad.to(dtype).float(), **tol)
    if rr is not None:
        torch.testing.assert_close(rk.grad[:, :c].float(), rr.grad.to(dtype).float(), **tol)
    if use_sb:
        torch.testing.assert_close(sbk.grad, sbr.grad, rtol=ptol['rtol'], atol=ptol['atol'] * h * w ** 0.5)
    scale = float(n * h * w) ** 0.5
    torch.testing.assert_close(bn.weight.grad, ref_bn.weight.grad, rtol=ptol['rtol'], atol=ptol['atol'] * scale)
    torch.testing.assert_close(bn.bias.grad, ref_bn.bias.grad, rtol=ptol['rtol'], atol=ptol['atol'] * scale)
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=1e-4)


@pytest.mark.parametrize('dtype,c', [(torch.bfloat16, 40), (torch.bfloat16, 32), (torch.float32, 12)])
def test_causal_pair_is_bit_exact(dtype, c):
    """stp3_causal_pair_fwd / _bwd (the operand of the causal (2,3,3) convolution) against the torch construction:
    zero frame + two concatenations, and autograd's slice gradient + accumulation."""
    from stp3_amd import ops
    b, t, h, w = 4, 3, 50, 40
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(b * t, c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b * t, 2 * c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    xk = x0.clone().requires_grad_()
    yk = ops.causal_pair(xk, t)
    yk.backward(gy)
    xr = x0.clone().requires_grad_()
    x5 = xr.view(b, t, c, h, w)
    prev = torch.cat([torch.zeros_like(x5[:, :1]), x5[:, :-1]], dim=1).view(b * t, c, h, w)
    yr = torch.cat([prev, xr], dim=1)
    yr.backward(gy)
    assert torch.equal(yk.detach(), yr.detach()) and torch.equal(xk.grad, xr.grad)
    wide = torch.randn(b * t, c + 8, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.causal_pair(wide[:, :c], t)[:, c:], wide[:, :c])      # channel slice read in place


@pytest.mark.parametrize('case', [(torch.bfloat16, 12, 64, 100, 100, 2), (torch.bfloat16, 72, 160, 14, 30, 2),
                                  (torch.float32, 3, 12, 25, 25, 2), (torch.bfloat16, 2, 8, 7, 5, 3),
                                  (torch.float32, 2, 4, 1, 6, 4)])
def test_bilinear_upsampling_matches_torch(case):
    """stp3_upsample_bilinear_fwd / _bwd against F.interpolate(mode='bilinear', align_corners=False) in float32 and its
    autograd (float32: 1e-6 relative; bf16: the float32 result rounded once, up to one bf16 ulp)."""
    import torch.nn.functional as F
    from stp3_amd import ops
    dtype, n, c, h, w, scale = case
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(n, c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, c, h * scale, w * scale, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    xk = x0.clone().requires_grad_()
    yk = ops.upsample_bilinear(xk, scale)
    yk.backward(gy)
    xr = x0.float().requires_grad_()
    yr = F.interpolate(xr, scale_factor=scale, mode='bilinear', align_corners=False)
    yr.backward(gy.float())
    assert yk.shape == yr.shape and yk.dtype == dtype and yk.is_contiguous(memory_format=torch.channels_last)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=8e-3, atol=1e-3)
    torch.testing.assert_close(yk.detach().float(), yr.detach().to(dtype).float(), **tol)
    gtol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=8e-3, atol=8e-3)
    torch.testing.assert_close(xk.grad.float(), xr.grad.to(dtype).float(), **gtol)


def test_upsampling_reads_a_concatenation_gradient_in_place():
    import torch.nn.functional as F
    from stp3_amd import ops
    g = torch.Generator().manual_seed(10)
    x0 = torch.randn(4, 16, 14, 30, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    skip = torch.randn(4, 8, 28, 60, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xk = x0.clone().requires_grad_()
    cat = torch.cat([skip, ops.upsample_bilinear(xk, 2)], dim=1)
    gc = torch.randn(cat.shape, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    cat.backward(gc)
    xr = x0.float().requires_grad_()
    F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=False).backward(gc[:, 8:].float())
    torch.testing.assert_close(xk.grad.float(), xr.grad.to(torch.bfloat16).float(), rtol=8e-3, atol=8e-3)


def _group_worker(rank, world, port, out):
    """Three sibling layers of one shard -- fused conv -> BatchNorm -> ReLU (bf16), BatchNorm + ReLU with a per-sample bias
    on 40-lane rows, a float32 BatchNorm -- as separate operators and as ONE exchange group; all-reduces counted."""
    import copy
    import torch.distributed as dist
    import torch.nn as nn
    from stp3_amd.layers import fused
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cl = torch.channels_last
    torch.manual_seed(5)
    conv = nn.Conv2d(16, 24, 3, padding=1, bias=False).cuda()
    bns = [nn.BatchNorm2d(24).cuda(), nn.BatchNorm2d(35).cuda(), nn.BatchNorm2d(8).cuda()]
    torch.manual_seed(100 + rank)
    x0 = torch.randn(2, 16, 20, 24).to(torch.bfloat16).cuda().contiguous(memory_format=cl)
    y1 = torch.zeros(2, 40, 20, 24).cuda().contiguous(memory_format=cl)
    y1[:, :35] = torch.randn(2, 35, 20, 24).cuda()
    y2 = torch.randn(2, 8, 20, 24).cuda().contiguous(memory_format=cl)
    sb = (torch.randn(2, 35) * 0.3).cuda()
    gys = [torch.randn(2, 24, 20, 24).to(torch.bfloat16).cuda().contiguous(memory_format=cl),
           torch.randn(2, 40, 20, 24).cuda().contiguous(memory_format=cl), torch.randn(2, 8, 20, 24).cuda().contiguous(memory_format=cl)]
    calls = {'n': 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    dist.all_reduce = counting
    results = []
    for grouped in (False, True):
        cv, b0, b1, b2 = copy.deepcopy(conv), *[copy.deepcopy(b) for b in bns]
        x, a1, a2, s1 = (t.clone().requires_grad_() for t in (x0, y1, y2, sb))
        calls['n'] = 0
        members = [fused.conv_bn_act_member(x, cv, b0, fused.ACT_RELU), dict(bn=b1, x=a1, act=fused.ACT_RELU, sbias=s1),
                   dict(bn=b2, x=a2)]
        assert not isinstance(members[0], dict)
        outs = fused.bn_act_group(members) if grouped else [fused._run_member(m) for m in members]
        fwd = calls['n']
        torch.autograd.backward(outs, gys)
        grads = [x.grad, cv.weight.grad, b0.weight.grad, b0.bias.grad, a1.grad, s1.grad, b1.weight.grad, b1.bias.grad, a2.grad,
                 b2.weight.grad, b2.bias.grad, b0.running_var, b1.running_mean, b2.running_var]
        results.append((fwd, calls['n'] - fwd, [o.detach().float().cpu() for o in outs], [g.detach().float().cpu() for g in grads]))
    dist.all_reduce = real
    (f0, b0c, o0, g0), (f1, b1c, o1, g1) = results
    out[rank] = {'separate': (f0, b0c), 'grouped': (f1, b1c), 'outputs_equal': all(torch.equal(a, b) for a, b in zip(o0, o1)),
                 'grads_equal': all(torch.equal(a, b) for a, b in zip(g0, g1))}
    dist.destroy_process_group()


def test_sibling_layers_share_one_exchange_two_ranks_one_gpu():
    """ops_fused._ExchangeGroup on the MI355X (two gloo ranks on cuda:0): the same bits as the separate operators, one
    all-reduce per pass instead of three."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_group_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in (0, 1):
        assert out[rank] == {'separate': (3, 3), 'grouped': (1, 1), 'outputs_equal': True, 'grads_equal': True}, out[rank]
