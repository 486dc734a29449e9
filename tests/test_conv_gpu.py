"""GPU parity of the bf16 MFMA implicit-GEMM convolution (through the C ABI) against a float32 torch
convolution of the same bf16-rounded operands.

Tolerances: float32 output rtol 2e-3 / atol 2e-3 (float32 accumulation in a different order over up to
K = 7*7*64 products of O(1) values); bf16 output additionally carries one bf16 rounding (2^-8 relative):
rtol 1e-2 / atol 2e-2.  Gradients: input gradient through the same kernel (same tolerances); weight gradient through the MFMA
weight-gradient kernel (float32 accumulation over up to ~1e5 pixels of bf16 products): rtol 2e-2 and an
absolute term of 2e-2 of the largest entry (bf16 rounding of dy and x in the reference is identical, the
difference is summation order); Cout = 2 (the heads) is zero-padded to 8 channels on the way.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil, bias, sliced
    (3, 24, 28, 60, 144, 1, 1, 0, 1, False, False),     # MBConv expand (K = 24 < one k-step)
    (2, 144, 28, 60, 32, 1, 1, 0, 1, False, False),     # MBConv project
    (2, 216, 28, 60, 64, 3, 1, 1, 1, False, False),     # UpsamplingConcat first 3x3
    (1, 64, 50, 50, 64, 7, 2, 3, 1, False, False),      # decoder stem 7x7 / 2
    (2, 64, 40, 40, 128, 3, 1, 12, 12, False, False),   # ASPP dilated 3x3
    (2, 64, 33, 17, 2, 1, 1, 0, 1, True, False),        # head: 2 output channels + bias, ragged pixel count
    (2, 160, 14, 30, 64, 3, 1, 24, 24, False, False),   # dilation > map: only the centre row is ever in range
    (3, 40, 20, 20, 35, 3, 1, 1, 1, False, True),       # channel-sliced input view (35 -> 40 padded), odd Cout
    (1, 128, 25, 25, 256, 3, 2, 1, 1, False, False),    # ResNet stage transition, stride 2
    (2, 960, 14, 30, 160, 1, 1, 0, 1, False, False),    # widest trunk projection (K = 960)
    (3, 8, 57, 121, 48, 3, 2, 0, 1, False, False),      # trunk stem (3 -> 8 padded channels): taps folded in the weight gradient
    (2, 8, 30, 22, 24, 5, 1, 2, 1, True, False),        # Cin == 8 with 25 taps: two folded tap groups
    (2, 64, 30, 31, 128, 1, 2, 0, 1, False, False),     # ResNet downsample 1x1 / 2: three of the four input phases get no tap
]


# the shapes bench.py's step launches (BASELINE configs[2]: B = 4, T = 3 -> 12 BEV frames of 200 x 200, 72 camera images)
BENCH_CASES = [
    (12, 64, 200, 200, 64, 3, 1, 1, 1, False, False),    # decoder heads / ResNet layer1 3x3 64 -> 64 @200x200x12 (the work-horse)
    (12, 128, 200, 200, 128, 3, 1, 1, 1, False, False),  # temporal DeepLabHead 3x3 128 -> 128 (128-wide tiles)
    (72, 24, 112, 240, 144, 1, 1, 0, 1, False, False),   # trunk expand 1x1 24 -> 144 @112x240x72 (one short K-step)
    (12, 64, 200, 200, 64, 7, 2, 3, 1, False, False),    # decoder stem 7x7 / 2 (per-phase data gradient at this size)
    # short-contraction 1x1 layers at sizes that take the streaming kernel (pointwise_kernel): both channel-tile widths, a k
    # tail (56 = 3.5 steps), the project shape whose DATA GRADIENT is one of them (K = 24 -> 144 channels)
    (72, 32, 56, 120, 192, 1, 1, 0, 1, False, False), (72, 56, 28, 60, 336, 1, 1, 0, 1, False, False),
    (72, 112, 14, 30, 672, 1, 1, 0, 1, False, False), (12, 144, 112, 240, 24, 1, 1, 0, 1, False, False),
]


@pytest.mark.parametrize('case', CASES + BENCH_CASES)
def test_conv2d_forward_and_input_gradient(case):
    from stp3_amd import ops
    n, cin, h, w, cout, k, stride, pad, dil, use_bias, sliced = case
    g = torch.Generator().manual_seed(cin * 7 + cout)
    cs = cin + 8 if sliced else cin
    xfull = torch.randn(n, cs, h, w, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda() if use_bias else None
    xa = xfull.detach().clone().requires_grad_(True)
    wa = wgt.detach().clone().requires_grad_(True)
    xin = xa[:, :cin] if sliced else xa
    assert ops.conv2d_supported(xin, wa, stride)
    y32 = ops.conv2d(xin, wa, bias, stride, pad, dil, out_dtype=torch.float32)
    y16 = ops.conv2d(xin, wa, bias, stride, pad, dil)
    xb = xfull.float()[:, :cin].detach().clone().requires_grad_(True)
    wb = wgt.to(torch.bfloat16).float().detach().clone().requires_grad_(True)
    ref = F.conv2d(xb, wb, bias, stride, pad, dil)
    assert y32.shape == ref.shape and y16.dtype == torch.bfloat16
    torch.testing.assert_close(y32, ref, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(y16.float(), ref, rtol=1e-2, atol=2e-2)
    gy = torch.randn(ref.shape, generator=g).cuda().to(torch.bfloat16)
    y16.backward(gy)
    ref.backward(gy.float())
    torch.testing.assert_close(xa.grad.float()[:, :cin], xb.grad, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(wa.grad, wb.grad, rtol=2e-2, atol=2e-2 * float(wb.grad.abs().max()))
    if use_bias:
        assert wa.grad.shape == wgt.shape


@pytest.mark.parametrize('case', [(2, 64, 50, 50, 64, 7, 3), (2, 64, 31, 30, 128, 3, 1), (2, 64, 30, 31, 128, 1, 0),
                                  (1, 128, 25, 25, 256, 3, 1), (2, 16, 20, 21, 8, 5, 2), (2, 8, 57, 121, 48, 3, 0)])
def test_strided_data_gradient_per_phase(case):
    """``ops._strided_dgrad`` (what ``conv2d_data_grad`` takes for big strided layers: the decoder's 7x7 / 2 stem at the
    bench size) on its own, against autograd of F.conv2d and against the zero-stuffed route."""
    from stp3_amd import ops
    n, cin, h, w, cout, k, pad = case
    g = torch.Generator().manual_seed(k * 100 + cin)
    x = torch.randn(n, cin, h, w, generator=g).cuda().requires_grad_()
    wgt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda().to(torch.bfloat16)
    y = F.conv2d(x, wgt.float(), None, 2, pad)
    gy = torch.randn(y.shape, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    (want,) = torch.autograd.grad(y, x, gy.float())
    wb = wgt.contiguous(memory_format=torch.channels_last)
    wt = wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
    cache = {}
    got = ops._strided_dgrad(gy, wt, tuple(x.shape), 2, (pad, pad), cache)
    assert got is not None and got.shape == x.shape and len(cache) >= 1
    torch.testing.assert_close(got.float(), want, rtol=1e-2, atol=2e-2)
    again = ops._strided_dgrad(gy, wt, tuple(x.shape), 2, (pad, pad), cache)          # cached sub-kernels
    assert torch.equal(got, again)
    stuffed = ops.conv2d_data_grad(gy, wb, None, tuple(x.shape), 2, (pad, pad), (1, 1))   # small layer: zero-stuffed route
    torch.testing.assert_close(got.float(), stuffed.float(), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize('case', [CASES[3], CASES[4], CASES[5], CASES[7], CASES[8], CASES[10]] + BENCH_CASES)
def test_conv2d_float32_route_against_float64(case):
    """``ops.conv2d_f32`` (float32 tensors outside autocast: three bf16 terms per operand, six term products on the MFMA
    kernels, float32 accumulators) against the same convolution in FLOAT64 -- forward, data gradient, weight gradient,
    relative to the largest entry: float32 accuracy.  This is the route every float32 leg of the step-level parity
    tests takes, so those pin ``conv2d_igemm_kernel`` / ``conv2d_wgrad_kernel`` themselves.  (The float64 side of the bench
    shapes runs on 2 of the samples for the forward / data gradient: a float64 vendor convolution is slow.)"""
    from stp3_amd import ops
    from stp3_amd.layers import fused
    n, cin, h, w, cout, k, stride, pad, dil, use_bias, sliced = case
    g = torch.Generator().manual_seed(cin * 11 + cout)
    x = (torch.randn(n, cin, h, w, generator=g) * 2.0 + 0.3).cuda().requires_grad_(True)
    wgt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda().requires_grad_(True)
    bias = torch.randn(cout, generator=g).cuda().requires_grad_(True) if use_bias else None
    assert fused._use_mfma_f32(x, wgt, stride)
    y = fused.conv2d(x, wgt, bias, stride, pad, dil)                 # the dispatch the models use
    assert y.dtype == torch.float32
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    m = min(n, 2)
    xd, wd = x.detach()[:m].double().requires_grad_(True), wgt.detach().double().requires_grad_(True)
    bd = bias.detach().double().requires_grad_(True) if use_bias else None
    ref = F.conv2d(xd, wd, bd, stride, pad, dil)
    ref.backward(gy[:m].double())

    def close(a, b, tol):
        assert (a.double() - b).abs().max().item() <= tol * b.abs().max().item(), ((a.double() - b).abs().max().item(), b.abs().max().item())
    close(y[:m], ref.detach(), 2e-6)
    close(x.grad[:m], xd.grad, 2e-6)
    if m == n:
        close(wgt.grad, wd.grad, 5e-6)
        if use_bias:
            close(bias.grad, bd.grad, 5e-6)
    else:                                   # all n samples: the weight gradient sample by sample in float64
        dw = torch.zeros_like(wd)
        for i in range(n):
            dw += torch.nn.grad.conv2d_weight(x.detach()[i:i + 1].double(), wd.shape, gy[i:i + 1].double(), stride, pad, dil)
        close(wgt.grad, dw, 5e-6)


def test_conv2d_rejects_what_it_cannot_do():
    from stp3_amd import ops
    x = torch.randn(1, 3, 8, 8).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(4, 3, 3, 3).cuda()
    assert not ops.conv2d_supported(x, w, 1)                      # Cin % 8 != 0
    with pytest.raises(Exception):
        ops.conv2d(x, w, None, 1, 1, 1)
    with pytest.raises(Exception):
        ops.conv2d(x.cpu(), w.cpu(), None, 1, 1, 1)               # no CPU fallback


def test_conv2d_on_tensors_beyond_2g_elements():
    """The kernels index PIXELS with 32 bits and address elements with 64-bit pointer arithmetic: a 2.3 G-element input
    (3 x 80 x 3072 x 3072, 4.5 GB of bf16 -- the expanded tensors of BASELINE configs[4] are of that order) through the
    forward, the data gradient and the weight gradient, checked on the crops where the offsets are largest and, for the
    weight gradient, against torch on the whole tensor."""
    from stp3_amd import ops
    n, cin, h, w, cout = 3, 80, 3072, 3072, 16
    assert n * cin * h * w > 2 ** 31
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(n, cin, h, w, device='cuda', dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * 0.1).to(torch.bfloat16).float().requires_grad_(True)
    xa = x.detach().requires_grad_(True)
    assert ops.conv2d_supported(xa, wgt, 1)
    y = ops.conv2d(xa, wgt, None, 1, 1, 1, out_dtype=torch.bfloat16)
    gy = torch.randn(y.shape, device='cuda', dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    for (i, r0, c0) in ((n - 1, h - 64, w - 64), (n - 1, 0, w - 64), (0, 0, 0), (1, 1500, 1500)):
        crop = x[i:i + 1, :, max(r0 - 1, 0):r0 + 65, max(c0 - 1, 0):c0 + 65].float()
        ref = F.conv2d(crop, wgt.detach(), None, 1, 1)
        dr, dc = (1 if r0 > 0 else 0), (1 if c0 > 0 else 0)
        inner = ref[:, :, dr:dr + 62, dc:dc + 62]                    # away from the crop's own zero padding
        got = y[i:i + 1, :, r0 + (0 if r0 else 0):r0 + 62, c0:c0 + 62].float()
        torch.testing.assert_close(got, inner, rtol=1e-2, atol=2e-2)
        # data gradient: correlation of dy with the flipped kernel, same crop logic
        gcrop = gy[i:i + 1, :, max(r0 - 1, 0):r0 + 65, max(c0 - 1, 0):c0 + 65].float()
        dref = F.conv_transpose2d(gcrop, wgt.detach(), None, 1, 1)[:, :, dr:dr + 62, dc:dc + 62]
        torch.testing.assert_close(xa.grad[i:i + 1, :, r0:r0 + 62, c0:c0 + 62].float(), dref, rtol=1e-2, atol=2e-2)
    # weight gradient against torch's own over the whole tensor (float32 accumulation over 28 M pixels of bf16 products)
    dw_ref = torch.zeros_like(wgt)
    for i in range(n):                                               # one image at a time: keeps the reference within 2^31 too
        dw_ref += torch.nn.grad.conv2d_weight(x[i:i + 1].float(), wgt.shape, gy[i:i + 1].float(), 1, 1)
    scale = dw_ref.abs().max()
    assert (wgt.grad - dw_ref).abs().max() <= 2e-2 * scale, ((wgt.grad - dw_ref).abs().max(), scale)
