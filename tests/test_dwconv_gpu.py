"""GPU: hand-written depthwise-conv kernels (forward / data grad / weight grad, through the C ABI)
against a float32 torch reference of the same op on the CPU (F.conv2d with groups == channels,
explicit "static same" padding) -- this is a floating-point kernel, tolerance stated per dtype:
  float32: rtol 1e-4 / atol 1e-5 (fp32 FMA accumulation, different order)
  bfloat16 I/O, fp32 accumulate: outputs compared after the same bf16 rounding, rtol 2e-2 / atol 2e-2
"""
import pytest
import torch
import torch.nn.functional as F

from tests import helpers as H

pytestmark = pytest.mark.gpu

# (N, C, H, W, K, stride, pad(l,r,t,b)) -- every MBConv depthwise shape class of the B4 trunk + edge cases
CASES = [
    (2, 48, 112, 240, 3, 1, (1, 1, 1, 1)),
    (2, 144, 112, 240, 3, 2, (0, 1, 0, 1)),     # asymmetric static-same padding
    (2, 192, 56, 120, 5, 2, (2, 2, 2, 2)),
    (3, 336, 28, 60, 5, 1, (2, 2, 2, 2)),
    (2, 672, 14, 30, 3, 1, (1, 1, 1, 1)),
    (2, 960, 14, 30, 5, 1, (2, 2, 2, 2)),
    (1, 8, 5, 7, 3, 2, (0, 1, 0, 1)),           # tiny / ragged
    (1, 16, 1, 1, 5, 1, (2, 2, 2, 2)),          # 1x1 map: only the centre tap is ever in range
    (3, 64, 50, 47, 7, 1, (3, 3, 3, 3)),        # the 7x7 layer of the prediction stage's ConvNeXt blocks
    (1, 8, 3, 4, 7, 1, (3, 3, 3, 3)),           # ... on a map smaller than the kernel
]


def _ref(x, w, stride, pad, dy):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv2d(F.pad(x, pad), w, None, stride, 0, 1, x.shape[1])
    y.backward(dy)
    return y.detach(), x.grad, w.grad


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_depthwise_conv_matches_torch_cpu(case, dtype):
    from stp3_amd import ops
    n, c, h, w_, k, s, pad = case
    x = H.det_tensor((n, c, h, w_), 3)
    wt = H.det_tensor((c, 1, k, k), 5, 0.5)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()                      # same rounded inputs on both sides
    ho = (h + pad[2] + pad[3] - k) // s + 1
    wo = (w_ + pad[0] + pad[1] - k) // s + 1
    dy = H.det_tensor((n, c, ho, wo), 7)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y_ref, dx_ref, dw_ref = _ref(x, wt, s, pad, dy)

    xg = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = wt.cuda().requires_grad_(True)
    y = ops.depthwise_conv2d(xg, wg, s, pad)
    assert y.shape == (n, c, ho, wo) and y.dtype == dtype
    y.backward(dy.to(dtype).cuda())
    if dtype == torch.float32:
        tol = dict(rtol=1e-4, atol=1e-5)
        wtol = dict(rtol=1e-4, atol=1e-4)
    else:
        tol = dict(rtol=2e-2, atol=2e-2)
        wtol = dict(rtol=1e-3, atol=1e-3 * max(1.0, (n * ho * wo) ** 0.5))   # fp32 accumulation of bf16 products
    torch.testing.assert_close(y.float().cpu(), y_ref, **tol)
    torch.testing.assert_close(xg.grad.float().cpu(), dx_ref, **tol)
    torch.testing.assert_close(wg.grad.float().cpu(), dw_ref, **wtol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_depthwise_7x7_with_bias_matches_torch_cpu(dtype):
    """stp3_dwconv2d_fwd_bias: nn.Conv2d(dim, dim, 7, padding=3, groups=dim) of stp3/layers/convolutions.py:318 -- output,
    input gradient, weight gradient and the bias gradient (per-channel sum of dy)."""
    from stp3_amd import ops
    n, c, h, w_ = 2, 64, 40, 36
    x = H.det_tensor((n, c, h, w_), 11)
    wt, b = H.det_tensor((c, 1, 7, 7), 12, 0.3), H.det_tensor((c,), 13, 0.5)
    dy = H.det_tensor((n, c, h, w_), 14)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, 1, 3, 1, c)
    yr.backward(dy)
    xg = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.depthwise_conv2d(xg, wg, 1, (3, 3, 3, 3), bias=bg)
    y.backward(dy.to(dtype).cuda())
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    wtol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-3, atol=1e-3 * (n * h * w_) ** 0.5)
    torch.testing.assert_close(y.float().cpu(), yr.detach(), **tol)
    torch.testing.assert_close(xg.grad.float().cpu(), xr.grad, **tol)
    torch.testing.assert_close(wg.grad.float().cpu(), wr.grad, **wtol)
    torch.testing.assert_close(bg.grad.float().cpu(), br.grad, **wtol)


def test_weight_gradient_is_bit_reproducible():
    from stp3_amd import ops
    x = H.det_tensor((4, 192, 56, 120), 1).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w = H.det_tensor((192, 1, 5, 5), 2, 0.5).cuda().requires_grad_(True)
    grads = []
    for _ in range(2):
        w.grad = None
        y = ops.depthwise_conv2d(x, w, 2, (2, 2, 2, 2))
        y.backward(torch.ones_like(y))
        grads.append(w.grad.clone())
    assert torch.equal(grads[0], grads[1])
