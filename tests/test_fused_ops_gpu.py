"""GPU parity of the fused operators of the training step: the BatchNorm statistics in the convolution epilogue and the
conv -> BN -> activation operator built on them, the squeeze-excite block and its gate MLP, the bf16 weight shadows, and
the clip + Adam kernels.

Tolerances as in test_conv_gpu.py / test_bnact_gpu.py (bf16 outputs: rtol 1e-2 / atol 2e-2; statistics: float32
sums of bf16-rounded values, rtol 1e-4 against the same sums computed by torch from the kernel's own output)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil, bias
    (3, 24, 28, 60, 144, 1, 1, 0, 1, False),
    (2, 144, 28, 60, 32, 1, 1, 0, 1, False),
    (2, 216, 28, 60, 64, 3, 1, 1, 1, False),
    (1, 64, 50, 50, 64, 7, 2, 3, 1, False),
    (2, 64, 40, 40, 128, 3, 1, 12, 12, False),
    (2, 64, 33, 17, 2, 1, 1, 0, 1, True),
    (3, 40, 20, 20, 35, 3, 1, 1, 1, False),
    (2, 960, 14, 30, 160, 1, 1, 0, 1, False),
]


@pytest.mark.parametrize('case', CASES)
def test_conv_forward_and_statistics_epilogue(case):
    from stp3_amd import ops, ops_fused
    n, cin, h, w, cout, k, stride, pad, dil, use_bias = case
    g = torch.Generator().manual_seed(cin + 3 * cout)
    x = torch.randn(n, cin, h, w, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    wb = wgt.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, generator=g).cuda() if use_bias else None
    sums = torch.empty(2, cout, dtype=torch.float32, device='cuda')
    y = ops_fused.conv2d_v2(x, wb, bias, stride, (pad, pad), (dil, dil), sums_ptr=sums.data_ptr())
    y0 = ops_fused.conv2d_v2(x, wb, bias, stride, (pad, pad), (dil, dil))
    ref = F.conv2d(x.float(), wb.float(), bias, stride, pad, dil)
    torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=2e-2)
    assert torch.equal(y, y0)                                              # the statistics epilogue does not change y
    yf = y.float()
    torch.testing.assert_close(sums[0], yf.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(sums[1], (yf * yf).sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize('act,res_mode', [('relu', 'none'), ('swish', 'none'), ('none', 'after'), ('relu', 'before')])
def test_fused_conv_bn_act_matches_the_two_operators(act, res_mode):
    from stp3_amd import ops, ops_fused
    from stp3_amd.layers import fused
    act_id = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'swish': ops.ACT_SWISH}[act]
    rm = {'none': ops.RES_NONE, 'before': ops.RES_BEFORE_ACT, 'after': ops.RES_AFTER_ACT}[res_mode]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 64, 24, 20, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn(3, 96, 24, 20, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(3, 96, 24, 20, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    conv_a, conv_b = nn.Conv2d(64, 96, 3, padding=1, bias=False).cuda(), nn.Conv2d(64, 96, 3, padding=1, bias=False).cuda()
    conv_b.load_state_dict(conv_a.state_dict())
    bn_a, bn_b = nn.BatchNorm2d(96).cuda(), nn.BatchNorm2d(96).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ra = res.clone().requires_grad_(True) if rm else None
    rb = res.clone().requires_grad_(True) if rm else None
    ya = ops_fused.conv_bn_act(xa, conv_a.weight, None, bn_a, act_id, ra, rm, 1, 1, 1)
    yb = fused.bn_act(bn_b, ops.conv2d(xb, conv_b.weight, None, 1, 1, 1), act_id, res=rb, res_mode=rm)
    torch.testing.assert_close(ya.float(), yb.float(), rtol=2e-2, atol=2e-2)
    ya.backward(gy)
    yb.backward(gy)
    torch.testing.assert_close(xa.grad.float(), xb.grad.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(conv_a.weight.grad, conv_b.weight.grad, rtol=2e-2, atol=2e-2 * float(conv_b.weight.grad.abs().max()))
    torch.testing.assert_close(bn_a.weight.grad, bn_b.weight.grad, rtol=2e-2, atol=5e-2)
    torch.testing.assert_close(bn_a.bias.grad, bn_b.bias.grad, rtol=2e-2, atol=5e-2)
    torch.testing.assert_close(bn_a.running_var, bn_b.running_var, rtol=1e-3, atol=1e-4)
    if rm:
        torch.testing.assert_close(ra.grad.float(), rb.grad.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_se_block_matches_torch(dtype):
    from stp3_amd import ops_fused
    g = torch.Generator().manual_seed(5)
    n, c, sq, h, w = 5, 144, 6, 28, 30
    x = torch.randn(n, c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, c, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    red_a, exp_a = nn.Conv2d(c, sq, 1).cuda(), nn.Conv2d(sq, c, 1).cuda()
    red_b, exp_b = nn.Conv2d(c, sq, 1).cuda(), nn.Conv2d(sq, c, 1).cuda()
    red_b.load_state_dict(red_a.state_dict())
    exp_b.load_state_dict(exp_a.state_dict())
    xa = x.clone().requires_grad_(True)
    xb = x.float().clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):                 # as in the training step: must not leak into the op
        ya = ops_fused.se_block(xa, red_a, exp_a)
    s = xb.mean((2, 3), keepdim=True)
    yb = torch.sigmoid(exp_b(F.silu(red_b(s)))) * xb
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(ya.float(), yb, **tol)
    ya.backward(gy)
    yb.backward(gy.float())
    torch.testing.assert_close(xa.grad.float(), xb.grad, **tol)
    ptol = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    for pa, pb in zip(list(red_a.parameters()) + list(exp_a.parameters()), list(red_b.parameters()) + list(exp_b.parameters())):
        torch.testing.assert_close(pa.grad, pb.grad, **ptol)


def test_fused_project_conv_with_drop_connect_and_skip():
    """The MBConv tail: 1x1 conv (thin channels) -> BN -> * drop-connect scale -> + skip, fused vs the two operators."""
    from stp3_amd import ops, ops_fused
    from stp3_amd.layers import fused
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 144, 20, 24, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    skip = torch.randn(4, 32, 20, 24, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, 32, 20, 24, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    oscale = torch.tensor([1.25, 0.0, 1.25, 1.25]).cuda()
    conv_a, conv_b = nn.Conv2d(144, 32, 1, bias=False).cuda(), nn.Conv2d(144, 32, 1, bias=False).cuda()
    conv_b.load_state_dict(conv_a.state_dict())
    bn_a, bn_b = nn.BatchNorm2d(32).cuda(), nn.BatchNorm2d(32).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    sa, sb = skip.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    ya = ops_fused.conv_bn_act(xa, conv_a.weight, None, bn_a, ops.ACT_NONE, sa, ops.RES_AFTER_ACT, group=False, oscale=oscale)
    yb = fused.bn_act(bn_b, ops.conv2d(xb, conv_b.weight), ops.ACT_NONE, res=sb, res_mode=ops.RES_AFTER_ACT, oscale=oscale)
    torch.testing.assert_close(ya.float(), yb.float(), rtol=2e-2, atol=2e-2)
    ya.backward(gy)
    yb.backward(gy)
    torch.testing.assert_close(xa.grad.float(), xb.grad.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(sa.grad.float(), sb.grad.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(conv_a.weight.grad, conv_b.weight.grad, rtol=2e-2, atol=2e-2 * float(conv_b.weight.grad.abs().max()))
    torch.testing.assert_close(bn_a.weight.grad, bn_b.weight.grad, rtol=2e-2, atol=5e-2)


def test_weight_shadows_match_the_torch_built_copies():
    """stp3_conv2d_prep_weights: one launch rewrites the bf16 forward / flipped copies of all
    registered weights; bit-identical with the per-layer torch operators of the default path."""
    from stp3_amd import ops
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 3), (144, 24, 1, 1), (35, 70, 3, 3), (64, 64, 7, 7), (2, 64, 1, 1), (160, 160, 3, 3)]
    weights = []
    for k, shp in enumerate(shapes):
        w = torch.randn(*shp, device='cuda')
        if k % 2 == 0:
            w = w.contiguous(memory_format=torch.channels_last)
        weights.append(nn.Parameter(w))
    sh = ops._WeightShadows(weights[0].device)
    for w in weights:
        sh.register(w)

    def check():
        for w in weights:
            ent = sh.lookup(w)
            ref_b = w.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ref_t = ref_b.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            assert torch.equal(ent['wb'], ref_b) and torch.equal(ent['wt'], ref_t)
            assert ent['wb'].permute(0, 2, 3, 1).is_contiguous() and ent['wt'].permute(0, 2, 3, 1).is_contiguous()

    check()
    with torch.no_grad():
        for w in weights:
            w.data.mul_(-0.37)
    sh.refresh()
    check()
    # and a convolution fed from the shadows equals one fed from the torch-built copies
    x = torch.randn(2, 64, 20, 28, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ent = sh.lookup(weights[0])
    y_shadow = ops._conv2d_launch(x, ent['wb'], None, 1, (1, 1), (1, 1), torch.bfloat16)
    ref_b = weights[0].detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert torch.equal(y_shadow, ops._conv2d_launch(x, ref_b, None, 1, (1, 1), (1, 1), torch.bfloat16))


def test_fused_clip_adam_matches_the_torch_operator_path(monkeypatch):
    """stp3_optim_clip_adam: one update from identical state == clip_grad_norm_ + FlatAdam.step."""
    from stp3_amd import parallel
    from stp3_amd.parallel import FlatAdam, GradientBuckets

    def make():
        torch.manual_seed(5)
        return nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.Conv2d(64, 64, 3, padding=1),
                             nn.Flatten(), nn.Linear(64 * 6 * 6, 500), nn.ReLU(), nn.Linear(500, 5)).cuda()

    ref_model, fus_model = make(), make()
    ref_b, fus_b = GradientBuckets(ref_model, bucket_bytes=1 << 20), GradientBuckets(fus_model, bucket_bytes=1 << 20, gather=False)
    assert len(ref_b.buckets) >= 3
    ref_opt = FlatAdam(ref_b, lr=1e-2, weight_decay=1e-3)
    fus_opt = FlatAdam(fus_b, lr=1e-2, weight_decay=1e-3)
    g = torch.Generator().manual_seed(2)
    for it in range(3):
        x = torch.randn(8, 3, 6, 6, generator=g).cuda()
        ref_b.zero_grad()
        ref_model(x).square().mean().backward()
        ref_b.finish()
        with torch.no_grad():
            for k in range(len(ref_b.buckets)):
                fus_b.buckets[k][0].copy_(ref_b.buckets[k][0])
                fus_b.flat_params[k].copy_(ref_b.flat_params[k])
                fus_opt.exp_avg[k].copy_(ref_opt.exp_avg[k])
                fus_opt.exp_avg_sq[k].copy_(ref_opt.exp_avg_sq[k])
            fus_opt.step_t.copy_(ref_opt.step_t)
        max_norm = 0.05 if it != 1 else 1e9
        monkeypatch.setattr(parallel, 'FUSED_ADAM', False)          # the torch-operator path (what CPU buckets take)
        n_ref = float(ref_opt.clip_and_step(max_norm))
        monkeypatch.setattr(parallel, 'FUSED_ADAM', True)
        n_fus = float(fus_opt.clip_and_step(max_norm))
        assert abs(n_fus - n_ref) <= 1e-5 * n_ref and fus_opt.step_count == ref_opt.step_count == it + 1
        for k in range(len(ref_b.buckets)):
            torch.testing.assert_close(fus_b.buckets[k][0], ref_b.buckets[k][0], rtol=5e-5, atol=1e-10)
            torch.testing.assert_close(fus_opt.exp_avg[k], ref_opt.exp_avg[k], rtol=5e-5, atol=1e-10)
            torch.testing.assert_close(fus_opt.exp_avg_sq[k], ref_opt.exp_avg_sq[k], rtol=5e-5, atol=1e-12)
            torch.testing.assert_close(fus_b.flat_params[k], ref_b.flat_params[k], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_se_mlp_kernels_match_the_torch_operator_mlp(monkeypatch, dtype):
    """stp3_se_mlp_fwd / _bwd: the squeeze-excite block with the gate MLP as single launches == the
    same block with the MLP written in torch operators (float32 in both), forward and all five gradients."""
    from stp3_amd import ops_fused
    torch.manual_seed(0)
    # (S a multiple of 4: the batched kernels -- 960 / 40 and 672 / 28 are the trunk's slow ones; 6 and 14: the general kernels;
    # 1100 channels: two chunks per thread, the second ragged)
    for n, c, s, hh, ww in [(72, 144, 6, 14, 30), (72, 960, 40, 7, 15), (72, 672, 28, 7, 15), (3, 48, 12, 5, 9), (5, 336, 14, 7, 9),
                            (4, 1100, 8, 3, 5)]:
        x0 = torch.randn(n, c, hh, ww, device='cuda').to(dtype).contiguous(memory_format=torch.channels_last)
        params0 = [torch.randn(s, c, 1, 1, device='cuda') * 0.1, torch.randn(s, device='cuda') * 0.1,
                   torch.randn(c, s, 1, 1, device='cuda') * 0.1, torch.randn(c, device='cuda') * 0.1]
        gy = torch.randn_like(x0)
        results = []
        for flag in (False, True):
            monkeypatch.setattr(ops_fused, '_SE_MLP', flag)
            x = x0.clone().requires_grad_()
            params = [p.clone().requires_grad_() for p in params0]
            y = ops_fused._SeBlock.apply(x, *params)
            y.backward(gy)
            results.append([y.detach().float()] + [t.grad.float() for t in [x] + params])
        tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
        for a, b in zip(*results):
            torch.testing.assert_close(b, a, **tol)


def _conv_bn_sync_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    from stp3_amd import ops, ops_fused
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)     # gloo moves CUDA tensors through the host
    torch.cuda.set_device(0)
    x, gy, conv, bn = _conv_bn_case()
    sl = slice(2 * rank, 2 * rank + 2)
    xa = x[sl].detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops_fused.conv_bn_act(xa, conv.weight, None, bn, ops.ACT_SWISH, None, ops.RES_NONE, 1, 1, 1)
    y.backward(gy[sl].contiguous(memory_format=torch.channels_last))
    out[rank] = (y.detach().float().cpu(), xa.grad.float().cpu(), conv.weight.grad.cpu(), bn.weight.grad.cpu(),
                 bn.running_var.cpu())
    dist.destroy_process_group()


def _conv_bn_case():
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 64, 24, 20, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, 96, 24, 20, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 96, 3, padding=1, bias=False).cuda()
    bn = nn.BatchNorm2d(96).cuda()
    return x, gy, conv, bn


def test_fused_conv_bn_act_cross_replica_statistics_two_ranks_one_gpu():
    """The fused conv -> BatchNorm operator with the statistics of its epilogue all-reduced over two ranks (half the
    batch each, both on cuda:0) == one rank with the whole batch: outputs, input gradient, summed weight gradients,
    running statistics (reference recipe: sync_batchnorm, /root/reference/train.py:47)."""
    import socket
    import torch.multiprocessing as mp
    from stp3_amd import ops, ops_fused
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_conv_bn_sync_worker, args=(2, port, out), nprocs=2, join=True)
    x, gy, conv, bn = _conv_bn_case()
    xa = x.detach().clone().requires_grad_(True)
    y = ops_fused.conv_bn_act(xa, conv.weight, None, bn, ops.ACT_SWISH, None, ops.RES_NONE, 1, 1, 1)
    y.backward(gy)
    torch.testing.assert_close(torch.cat([out[0][0], out[1][0]]), y.detach().float().cpu(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(torch.cat([out[0][1], out[1][1]]), xa.grad.float().cpu(), rtol=2e-2, atol=2e-2)
    wg = conv.weight.grad.cpu()
    torch.testing.assert_close(out[0][2] + out[1][2], wg, rtol=2e-2, atol=2e-2 * float(wg.abs().max()))
    torch.testing.assert_close(out[0][3] + out[1][3], bn.weight.grad.cpu(), rtol=2e-2, atol=5e-2)
    torch.testing.assert_close(out[0][4], bn.running_var.cpu(), rtol=1e-3, atol=1e-4)


# (c, kernel, stride, n, h, w, canonical image size of the "static same" padding): the trunk's depthwise layers at
# batch 2 x 6 cameras, plus a small odd-sized one
MBCONV_MID = [(48, 3, 1, 12, 112, 240, 190), (144, 3, 2, 12, 112, 240, 190), (192, 3, 1, 12, 56, 120, 95),
              (192, 5, 2, 12, 56, 120, 95), (336, 5, 1, 12, 28, 60, 48), (672, 3, 1, 12, 14, 30, 24),
              (960, 5, 1, 12, 14, 30, 24), (24, 3, 1, 3, 9, 13, 8)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', MBCONV_MID)
def test_mbconv_middle_operator_matches_torch_float32(case, dtype):
    """ops_fused.dw_bn_se -- depthwise (statistics in its epilogue) -> BN1 -> swish -> squeeze-excite with swish(BN1(.))
    applied on load and never written, one-pass backward reductions -- against float32 torch autograd on the same
    (bf16-representable) data at the trunk's shapes.  float32: rtol 1e-4 everywhere.  bf16: what is stored in bf16 (the
    output, the input gradient, and the depthwise weight gradient that is computed from the bf16 dE2) to 2e-2; every
    other gradient is a float32 reduction of exactly representable inputs."""
    from stp3_amd import ops_fused
    from stp3_amd.models.efficientnet import StaticSamePadConv2d
    c, k, stride, n, h, w, img = case
    cl = torch.channels_last
    g = torch.Generator().manual_seed(c + k)
    x0 = torch.randn(n, c, h, w, generator=g).to(dtype).cuda().contiguous(memory_format=cl)
    s = max(1, c // 24)
    res = []
    for mode in ('fused', 'torch'):
        dw = StaticSamePadConv2d(c, c, k, img, stride=stride, groups=c)
        bn = nn.BatchNorm2d(c, momentum=0.01, eps=1e-3)
        r1, r2 = StaticSamePadConv2d(c, s, 1, 1, bias=True), StaticSamePadConv2d(s, c, 1, 1, bias=True)
        gp = torch.Generator().manual_seed(5)
        with torch.no_grad():
            dw.weight.copy_(torch.randn(dw.weight.shape, generator=gp) * 0.3)
            bn.weight.copy_(torch.rand(c, generator=gp) + 0.5); bn.bias.copy_(torch.randn(c, generator=gp) * 0.2)
            r1.weight.copy_(torch.randn(r1.weight.shape, generator=gp) * 0.1); r1.bias.copy_(torch.randn(s, generator=gp) * 0.2)
            r2.weight.copy_(torch.randn(r2.weight.shape, generator=gp) * 0.3); r2.bias.copy_(torch.randn(c, generator=gp) * 0.2)
        for m in (dw, bn, r1, r2):
            m.cuda()
        if mode == 'fused':
            x = x0.clone().requires_grad_()
            assert ops_fused.dw_bn_se_supported(x, dw, bn)
            y = ops_fused.dw_bn_se(x, dw, bn, r1, r2, group=False)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dtype).cuda().contiguous(memory_format=cl)
            y.backward(gy)
        else:
            x = x0.double().requires_grad_()
            for m in (dw, bn, r1, r2):
                m.double()
            e2 = F.conv2d(F.pad(x, dw._pad), dw.weight, None, stride, 0, 1, c)
            if dtype == torch.bfloat16:
                e2 = e2 + (e2.to(torch.bfloat16).double() - e2).detach()              # the kernel stores E2 in bf16
            sact = F.silu(bn(e2))
            gate = torch.sigmoid(F.linear(F.silu(F.linear(sact.mean((2, 3)), r1.weight.flatten(1), r1.bias)),
                                          r2.weight.flatten(1), r2.bias))
            y = sact * gate[:, :, None, None]
            y.backward(gy.double())
        res.append({'y': y.detach(), 'dx': x.grad, 'ddw': dw.weight.grad, 'dgamma': bn.weight.grad, 'dbeta': bn.bias.grad,
                    'dw1': r1.weight.grad, 'db1': r1.bias.grad, 'dw2': r2.weight.grad, 'db2': r2.bias.grad,
                    'rmean': bn.running_mean.clone(), 'rvar': bn.running_var.clone()})
    got, want = res
    stored = ('y', 'dx', 'ddw')
    for name in got:
        a, b = got[name].double(), want[name].double()
        err = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        tol = 1e-4 if dtype == torch.float32 else (2e-2 if name in stored else 2e-3)
        assert err <= tol, (name, err)


@pytest.mark.parametrize('shape', [(72, 24, 144, 112, 240), (72, 160, 960, 14, 30), (8, 32, 192, 56, 120), (3, 56, 336, 27, 61)])
def test_pointwise_conv_bn_act_without_the_convolution_output(shape):
    """ops_fused._PointwiseBnAct (the MBConv expand convolution -> BN0 -> swish with the expanded pre-activation tensor
    recomputed instead of stored) at the trunk's real shapes against the stored route ops_fused._ConvBnAct: same kernels,
    same rounding places -- outputs and running statistics bit-equal, gradients to float32 summation order; and both within bf16 accuracy of float32 torch on the same operands."""
    import torch.nn as nn
    import torch.nn.functional as F
    from stp3_amd import ops, ops_fused
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(cin + cout)
    conv = nn.Conv2d(cin, cout, 1, bias=False).cuda()
    x0 = torch.randn(n, cin, h, w, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, cout, h, w, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for recompute in (True, False):
        bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, cout))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, cout))
        conv.zero_grad()
        x = x0.clone().requires_grad_()
        assert ops_fused.pointwise_bn_act_supported(x, conv, bn)
        y = (ops_fused.pointwise_bn_act(x, conv, bn, ops.ACT_SWISH, group=False) if recompute
             else ops_fused.conv_bn_act(x, conv.weight, None, bn, ops.ACT_SWISH, group=False))
        y.backward(gy)
        ops.flush_batch_counters()
        res.append((y.detach(), x.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                    bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])
    # the input gradient: the two float32 sums of the BatchNorm backward are added in another order on the recomputing route
    # (per workgroup of the streaming kernel instead of per 128-pixel block), so single bf16 roundings of dE0 may flip
    assert ((a[1].float() - b[1].float()).norm() <= 1e-3 * b[1].float().norm())
    assert int(a[7]) == 1
    for i in (2, 3, 4):
        assert (a[i] - b[i]).abs().max() <= 2e-5 * b[i].abs().max() + 1e-7, i
    if n * h * w <= 200000:                                        # float32 torch on the same operands (the small shapes)
        xr = x0.float().requires_grad_()
        wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_()
        e0 = F.conv2d(xr, wr).to(torch.bfloat16).float()
        mean, var = e0.mean((0, 2, 3)), e0.var((0, 2, 3), unbiased=False)
        ref = F.silu((e0 - mean[None, :, None, None]) * torch.rsqrt(var + 1e-3)[None, :, None, None]
                     * torch.linspace(0.5, 1.5, cout).cuda()[None, :, None, None] + torch.linspace(-0.3, 0.3, cout).cuda()[None, :, None, None])
        torch.testing.assert_close(a[0].float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('channels,out_channels,pyramid', [(70, 64, True), (64, 64, True), (64, 64, False)])
def test_temporal_block_paths_written_into_one_buffer_bit_equal(monkeypatch, channels, out_channels, pyramid):
    """TemporalBlock (stp3/layers/temporal.py:426-489): its three paths write their BatchNorm outputs into the channel
    slices of the aggregation's operand (``out_slot`` of ``ops.bn_act`` + ``ops_fused.join_slices``) instead of being
    concatenated -- against the same block with the concatenation: outputs, input gradient and every parameter gradient
    bit for bit (same kernels on the same values; only where the rows land differs)."""
    from stp3_amd.layers import fused, temporal
    from stp3_amd.utils import to_channels_last

    def run(slots):
        monkeypatch.setattr(temporal, 'slot_ok', fused.slot_ok if slots else (lambda x: False))
        torch.manual_seed(4)
        extra_ch = channels - 64
        blk = to_channels_last(temporal.TemporalBlock(channels, out_channels, use_pyramid_pooling=pyramid,
                                                      pool_sizes=[(2, 40, 48)] if pyramid else None).cuda())
        blk.train()
        g = torch.Generator().manual_seed(8)
        x = torch.randn(2, 64, 3, 40, 48, generator=g).cuda().requires_grad_()
        extra = torch.randn(2, extra_ch, 3, generator=g).cuda() if extra_ch else None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = blk(x, extra) if extra is not None else blk(x)
        gy = torch.randn(y.shape, generator=g).cuda().to(y.dtype)
        y.backward(gy)
        return [y.detach(), x.grad] + [p.grad for p in blk.parameters()]

    cat, slot = run(False), run(True)
    assert len(cat) == len(slot) and float(cat[0].float().abs().max()) > 0
    for i, (a, b) in enumerate(zip(cat, slot)):
        assert a is not None and b is not None and torch.equal(a, b), i


def test_upsampling_concat_written_into_one_buffer_bit_equal(monkeypatch):
    """UpsamplingConcat (stp3/layers/convolutions.py:183-205): the up-sampling kernel writes into its channel slice of the
    3x3 convolution's operand and the skip is copied into the other, instead of ``torch.cat`` of the two -- against the
    concatenating form: output, both input gradients and every parameter gradient bit for bit."""
    from stp3_amd.layers import convolutions, fused
    from stp3_amd.utils import to_channels_last

    def run(slots):
        monkeypatch.setattr(fused, 'slot_ok', (lambda x: x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16) if slots
                            else (lambda x: False))
        torch.manual_seed(6)
        mod = to_channels_last(convolutions.UpsamplingConcat(160 + 56, 64).cuda())
        mod.train()
        g = torch.Generator().manual_seed(2)
        lo = torch.randn(3, 160, 14, 30, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
        skip = torch.randn(3, 56, 28, 60, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = mod(lo, skip)
        y.backward(torch.randn(y.shape, generator=g).cuda().to(y.dtype))
        return [y.detach(), lo.grad, skip.grad] + [p.grad for p in mod.parameters()]

    cat, slot = run(False), run(True)
    assert len(cat) == len(slot) and float(cat[0].float().abs().max()) > 0
    for i, (a, b) in enumerate(zip(cat, slot)):
        assert a is not None and b is not None and torch.equal(a, b), i


@pytest.mark.parametrize('m,k,n,bias', [(12, 64, 128, True), (16, 70, 23, False), (72, 160, 64, True), (12, 6, 35, False)])
def test_small_linear_matches_torch(m, k, n, bias):
    """stp3_linear_fwd / _bwd -- the 1x1 convolutions of the pooled descriptors (stp3/layers/convolutions.py:229-240,
    stp3/layers/temporal.py:380-424) as one launch each way -- against float64 torch, rtol 1e-5 (float32 sums of <= 160 terms)."""
    from stp3_amd import ops
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).cuda().requires_grad_()
    w = torch.randn(n, k, generator=g).cuda().requires_grad_()
    b = torch.randn(n, generator=g).cuda().requires_grad_() if bias else None
    assert ops.small_linear_supported(x, w, b)
    y = ops.small_linear(x, w, b)
    gy = torch.randn(m, n, generator=g).cuda()
    y.backward(gy)
    xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    br = b.detach().double().requires_grad_() if bias else None
    yr = F.linear(xr, wr, br)
    yr.backward(gy.double())
    for got, want in ((y, yr), (x.grad, xr.grad), (w.grad, wr.grad)) + (((b.grad, br.grad),) if bias else ()):
        torch.testing.assert_close(got.detach().double(), want.detach(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('shape', [(4, 32, 64, 64), (3, 112, 14, 30)])
def test_skip_gradient_added_in_the_data_gradient_kernel_bit_equal(monkeypatch, shape):
    """An MBConv block with an identity skip (stp3/models/encoder.py:57-97 drives efficientnet_pytorch's MBConvBlock): the
    skip's gradient travels from the project BatchNorm to the expand convolution through an ``ops.SkipCarrier`` and is added
    in the epilogue of the data-gradient kernel (stp3_conv2d_fwd_add) instead of by the autograd engine -- output, input
    gradient and every parameter gradient bit for bit against the engine's own addition; both expand routes (the recomputing
    streaming one for the big map, the stored one for the small map)."""
    from stp3_amd import _lib
    from stp3_amd.models import efficientnet as E
    from stp3_amd.utils import to_channels_last
    n, c, h, w = shape

    def run(on):
        monkeypatch.setattr(E, 'SKIP_GRADIENT_IN_DGRAD', on)
        torch.manual_seed(0)
        blk = to_channels_last(E.MBConvBlock(c, c, 3, 1, 6, 64).cuda())
        blk.train()
        g = torch.Generator().manual_seed(1)
        x0 = torch.randn(n, c, h, w, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
        calls = []
        # (the skip gradient is added by stp3_conv2d_fwd_add -- the data-gradient convolution -- or, where the expand layer's
        # BatchNorm-backward apply pass computes the data gradient itself, by stp3_conv2d_bn_bwd_apply_dx)
        for entry in ('stp3_conv2d_fwd_add', 'stp3_conv2d_bn_bwd_apply_dx'):
            real = getattr(_lib.lib(), entry)
            monkeypatch.setattr(_lib.lib(), entry, (lambda real: lambda *a: (calls.append(a[-3 if real.__name__.endswith('_dx') else 4]), real(*a))[1])(real),
                                raising=False)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = blk(x0 * 1.0, drop_connect_rate=0.0)
        y.backward(torch.randn(y.shape, generator=g).cuda().to(y.dtype))
        return [y.detach(), x0.grad] + [p.grad for p in blk.parameters()], sum(1 for add in calls if add)

    (plain, n_plain), (fused, n_fused) = run(False), run(True)
    assert n_plain == 0 and n_fused == 1                     # (calls that were handed a skip gradient)
    for i, (a, b) in enumerate(zip(plain, fused)):
        assert a is not None and b is not None and torch.equal(a, b), i


@pytest.mark.parametrize('cin,cout', [(24, 144), (32, 192)])
def test_expand_data_gradient_inside_the_apply_pass(monkeypatch, cin, cout):
    """stp3_conv2d_bn_bwd_apply_dx: the expand convolution's data gradient computed by the BatchNorm-backward apply pass of the
    recomputing route (from the gradient tile it holds in LDS) against the separate data-gradient convolution: every gradient
    of ``ops_fused.pointwise_bn_act`` (MBConv expand layer, stp3/models/encoder.py:57-97) within bf16 rounding of one another
    (rtol 1e-2; the two sum the same bf16 products in float32)."""
    from stp3_amd import ops_fused
    conv = nn.Conv2d(cin, cout, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(cout, momentum=0.01, eps=1e-3).cuda()

    def run(on):
        monkeypatch.setattr(ops_fused, 'EXPAND_DGRAD_IN_APPLY', on)
        conv.weight.grad = bn.weight.grad = bn.bias.grad = None
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, cin, 72, 80, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
        assert ops_fused.pointwise_bn_act_supported(x, conv, bn) and ops_fused.pointwise_bn_act_pays(x, conv)
        y = ops_fused.pointwise_bn_act(x, conv, bn, 2, group=False)
        y.backward(torch.randn(y.shape, generator=g).cuda().to(y.dtype))
        return [y.detach().float(), x.grad.float(), conv.weight.grad.float(), bn.weight.grad.float(), bn.bias.grad.float()]

    two, one = run(False), run(True)
    assert torch.equal(two[0], one[0]) and torch.equal(two[2], one[2]) and torch.equal(two[3], one[3])
    torch.testing.assert_close(one[1], two[1], rtol=1e-2, atol=1e-2 * float(two[1].abs().max()))
    print('dx bit-equal:', bool(torch.equal(one[1], two[1])))


def test_assembled_weights_equal_the_torch_built_ones(monkeypatch):
    """ops.ASSEMBLED_WEIGHTS: weights the model puts together from parameters (padded lanes and causal taps of the temporal
    block, stp3/layers/temporal.py:8-37 / stp3/models/temporal_model.py; ASPP's kept taps and split projection,
    stp3/layers/convolutions.py; the merged decoder heads, stp3/models/decoder.py:96-140) as shadows written piece by piece by
    stp3_conv2d_prep_weights, their gradients cut back into the bucket slices by ONE stp3_conv2d_scatter_weight_grads per
    backward pass -- against the same modules building the weights with torch.  Same kernels on the same bf16 operands: the
    loss and the flat gradient buckets are bit-equal, over two passes with a parameter update in between."""
    from stp3_amd import _lib, ops
    from stp3_amd.layers import temporal as T, convolutions as C, fused
    from stp3_amd.models import decoder as D
    from stp3_amd.parallel import GradientBuckets
    from stp3_amd.utils import to_channels_last
    gate = {'perceive_hdmap': False, 'predict_pedestrian': True, 'predict_instance': True, 'predict_future_flow': False, 'planning': False}

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.first = T.TemporalBlock(70, 64, use_pyramid_pooling=True, pool_sizes=[(2, 32, 48)])   # 64 planes + 6 constants
            self.second = T.TemporalBlock(64, 64)
            self.aspp = C.ASPP(64, [2, 36, 60], 32)            # 32 x 48 map: all taps, centre row only, centre tap only
            self.decoder = D.Decoder(32, 2, 2, 2, gate)

        def forward(self, x, extra):
            b, _, t = extra.shape
            x = self.second(self.first(x, extra))
            y = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], *x.shape[3:]).contiguous(memory_format=torch.channels_last)
            out = self.decoder(self.aspp(y).view(b, t, -1, *y.shape[2:]))
            return sum(v.float().square().mean() for v in out.values() if v is not None)

    lib = _lib.lib()
    counts = {}

    class Counting:
        def __getattr__(self, name):
            fn = getattr(lib, name)
            if name != 'stp3_conv2d_scatter_weight_grads':
                return fn

            def wrapped(*a):
                counts[name] = counts.get(name, 0) + 1
                return fn(*a)
            return wrapped

    monkeypatch.setattr(_lib, 'lib', lambda: Counting())

    def run(on):
        monkeypatch.setattr(ops, 'ASSEMBLED_WEIGHTS', on)
        torch.manual_seed(5)
        model = to_channels_last(Net()).cuda().train()
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        buckets = GradientBuckets(model)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(2, 64, 2, 32, 48, generator=g).cuda()
        extra = torch.randn(2, 6, 2, generator=g).cuda()
        losses, flats = [], []
        for _ in range(2):
            buckets.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                loss = model(x, extra)
            loss.backward()
            buckets.finish()
            fused.flush_batch_counters()
            losses.append(float(loss.detach()))
            flats.append(torch.cat([f.clone() for f, _ in buckets.buckets]))
            with torch.no_grad():
                for fp in buckets.flat_params:
                    fp.mul_(0.97)
            ops.invalidate_weight_cache()
        return losses, flats

    plain = run(False)
    assert not counts
    asm = run(True)
    assert counts == {'stp3_conv2d_scatter_weight_grads': 2}            # one launch per backward pass
    assert sum(len(t.assembled) for t in ops._SHADOW_TABLES.values()) >= 12
    assert asm[0] == plain[0], (asm[0], plain[0])
    for a, b in zip(asm[1], plain[1]):
        assert float(b.abs().max()) > 0
        assert torch.equal(a, b), float((a - b).abs().max() / b.abs().max())
