"""GPU: LayerNorm over the channels of every pixel (+ GELU) through the C ABI (stp3_layernorm_fwd / _bwd) against float64
torch autograd of `LayerNorm` + `nn.GELU()` as the reference composes them (stp3/layers/convolutions.py:283-307,
:347-380) -- floating point, tolerances: float32 rows 1e-5 relative (float32 arithmetic, another summation order), bf16
rows 8e-3 (one rounding of the result), parameter gradients 1e-4 (float32 accumulation over all pixels)."""
import pytest
import torch
import torch.nn.functional as F

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('shape', [(4, 64, 200, 200), (3, 32, 50, 47), (1, 128, 5, 3), (2, 256, 9, 8)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('gelu', [False, True])
def test_layernorm_matches_float64_autograd(shape, dtype, gelu):
    from stp3_amd import ops_pred
    n, c, h, w = shape
    x0 = (H.det_tensor(shape, 31) * 1.5 + 0.3).to(dtype)
    w0, b0 = H.det_tensor((c,), 32, 0.5) + 1.0, H.det_tensor((c,), 33, 0.3)
    gy = H.det_tensor(shape, 34).to(dtype)
    xr, wr, br = x0.double().requires_grad_(True), w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    yr = F.layer_norm(xr.permute(0, 2, 3, 1), (c,), wr, br, 1e-6).permute(0, 3, 1, 2)
    yr = F.gelu(yr) if gelu else yr
    yr.backward(gy.double())
    x = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = w0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
    assert ops_pred.layer_norm_supported(x, c)
    y = ops_pred.layer_norm_channels(x, wg, bg, 1e-6, ops_pred.ACT_GELU if gelu else ops_pred.ACT_NONE)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.cuda())
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert _rel(y.detach().float(), yr.detach()) <= tol
    assert _rel(x.grad.float(), xr.grad) <= tol
    assert _rel(wg.grad, wr.grad) <= 1e-4 and _rel(bg.grad, br.grad) <= 1e-4


def test_layernorm_is_bit_reproducible_and_takes_a_channel_slice():
    from stp3_amd import ops_pred
    full = H.det_tensor((2, 96, 40, 36), 41).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    x = full[:, 16:80].detach().requires_grad_(True)              # row stride 96, 64 channels
    wg, bg = (H.det_tensor((64,), 42, 0.5) + 1.0).cuda().requires_grad_(True), H.det_tensor((64,), 43, 0.3).cuda().requires_grad_(True)
    outs = []
    for _ in range(2):
        x.grad = wg.grad = bg.grad = None
        y = ops_pred.layer_norm_channels(x, wg, bg, 1e-6, ops_pred.ACT_GELU)
        y.backward(torch.ones_like(y))
        outs.append((y.detach().clone(), x.grad.clone(), wg.grad.clone(), bg.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    ref = F.gelu(F.layer_norm(x.detach().float().permute(0, 2, 3, 1), (64,), wg.detach(), bg.detach(), 1e-6)).permute(0, 3, 1, 2)
    assert _rel(outs[0][0].float(), ref) <= 8e-3


def test_unsupported_channel_counts_are_refused_not_miscomputed():
    from stp3_amd import _lib, ops_pred
    x = torch.zeros(1, 24, 4, 4, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not ops_pred.layer_norm_supported(x, 24)                # 3 lanes per row: not a power of two
    with pytest.raises(_lib.Stp3HipError):
        ops_pred.layer_norm_channels(x, None, None, 1e-6)
