"""Operators of the prediction stage on the HIP library: LayerNorm over the channels of every
pixel (+ GELU) -- ``stp3/layers/convolutions.py:283-307`` and the ``nn.GELU()`` behind it in ``Bottleblock`` (:347-380).
GPU only; the layers keep torch's operators for shapes the kernels do not take (``layer_norm_supported``)."""
import ctypes

import torch

from . import _lib, ops
from ._lib import check

ACT_NONE, ACT_GELU = _lib.ACT_NONE, _lib.ACT_GELU
_WS = {}


def _workspace(nbytes, device):
    key = ops._ws_key(device)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def layer_norm_supported(x, channels):
    """What stp3_layernorm_* take: a 4-D (N,C,H,W) GPU tensor in bf16 / float32 whose channel count is 16 bytes of elements
    times a power of two <= 64 (the prediction stage: 32 and 64 channels)."""
    if not (x.is_cuda and x.dim() == 4 and x.shape[1] == channels and x.dtype in (torch.bfloat16, torch.float32)):
        return False
    lanes, rem = divmod(channels, 8 if x.dtype == torch.bfloat16 else 4)
    return rem == 0 and 1 <= lanes <= 64 and lanes & (lanes - 1) == 0


def _dims(x, ldx, ldy, act, eps):
    n, c, h, w = x.shape
    return _lib.LayerNormDims(n * h * w, c, ldx, ldy, _lib.DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.DTYPE_F32,
                              int(act), float(eps))


class _LayerNormChannels(torch.autograd.Function):
    """x (N,C,H,W), channels-last memory -> act(LayerNorm over C) in x's type and layout; weight / bias (C)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act):
        ops._need_gpu(x)
        x, ldx = ops._rows_view(x)
        y = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        w32, b32 = ops._f32(weight), ops._f32(bias)
        dims = _dims(x, ldx, x.shape[1], act, eps)
        check(_lib.lib().stp3_layernorm_fwd(ctypes.byref(dims), ops._ptr(x), ops._opt_ptr(w32), ops._opt_ptr(b32),
                                            ops._ptr(y), ops._stream()), 'stp3_layernorm_fwd')
        ctx.save_for_backward(x, weight, bias)
        ctx.conf = (ldx, float(eps), int(act))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        ldx, eps, act = ctx.conf
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy, ldy = ops._rows_view(dy)
        w32, b32 = ops._f32(weight), ops._f32(bias)
        c = x.shape[1]
        # dx in x's own row layout (a channel slice keeps its stride); gradients of the parameters float32
        dx = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device) if ldx != c else \
            torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dims = _dims(x, ldx, ldy, act, eps)
        nbytes = ctypes.c_size_t()
        lib = _lib.lib()
        check(lib.stp3_layernorm_bwd_workspace(ctypes.byref(dims), ctypes.byref(nbytes)), 'stp3_layernorm_bwd_workspace')
        ws = _workspace(nbytes.value, x.device)
        need_w = weight is not None and ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        dg = torch.empty(c, dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty(c, dtype=torch.float32, device=x.device) if need_b else None
        check(lib.stp3_layernorm_bwd(ctypes.byref(dims), ops._ptr(dy), ops._ptr(x), ops._opt_ptr(w32), ops._opt_ptr(b32),
                                     ops._ptr(dx), ops._opt_ptr(dg), ops._opt_ptr(db), ops._ptr(ws),
                                     ctypes.c_size_t(nbytes.value), ops._stream()), 'stp3_layernorm_bwd')
        if need_w and dg.dtype != weight.dtype:
            dg = dg.to(weight.dtype)
        if need_b and db.dtype != bias.dtype:
            db = db.to(bias.dtype)
        return dx, dg, db, None, None


_LN_APPLY = ops._fast_apply(_LayerNormChannels)


def layer_norm_channels(x, weight, bias, eps, act=ACT_NONE):
    """LayerNorm over dim 1 of (N,C,H,W) (+ GELU) on the kernels.  Under autocast the tensor runs in the autocast type
    (arithmetic float32 inside the kernel, one rounding -- where torch computes float32 and the consumer rounds)."""
    if torch.is_autocast_enabled() and x.dtype != torch.get_autocast_dtype('cuda'):
        x = x.to(torch.get_autocast_dtype('cuda'))
    return _LN_APPLY(x, weight, bias, float(eps), int(act))
