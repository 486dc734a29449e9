"""Operators of the prediction stage on the HIP library: LayerNorm over the channels of every
pixel (+ GELU) -- ``stp3/layers/convolutions.py:283-307`` and the ``nn.GELU()`` behind it in ``Bottleblock`` (:347-380).
GPU only; the layers keep torch's operators for shapes the kernels do not take (``layer_norm_supported``)."""
import ctypes
import weakref

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
    times a power of two in 4 .. 64 (the prediction stage: 32 and 64 channels)."""
    if not (x.is_cuda and x.dim() == 4 and x.shape[1] == channels and x.dtype in (torch.bfloat16, torch.float32)):
        return False
    per = 8 if x.dtype == torch.bfloat16 else 4
    lanes, rem = divmod(channels, per)
    if not (rem == 0 and 4 <= lanes <= 64 and lanes & (lanes - 1) == 0):
        return False
    # rows as the kernel will see them (ops._rows_view keeps a channel slice of a channels-last tensor in place, anything it
    # cannot use becomes a fresh dense copy): 16-byte aligned, row stride a multiple of the vector width -- otherwise
    # stp3_layernorm_* answer STP3_EUNSUP and the caller keeps torch's operator
    n, c, h, w = x.shape
    if x.is_contiguous(memory_format=torch.channels_last):
        return x.data_ptr() % 16 == 0
    sn, sc, sh, sw = x.stride()
    if c > 1 and sc != 1:
        return True
    ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    in_place = ld >= c and (w == 1 or sw == ld) and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld)
    return not in_place or (ld % per == 0 and x.data_ptr() % 16 == 0)


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


# ----------------------------------------------------------------------------------------------
# Convolutional GRU cell: stp3/layers/temporal.py:42-56 (SpatialGRU.gru_cell), :118-145 (Dual_GRU.gru_cell_1 / _2)
# ----------------------------------------------------------------------------------------------
_GATE_WEIGHTS = {}        # (id(update weight), id(reset weight)) -> [stamp, merged bf16 weight, merged float32 bias]


def _merged_gate_weights(conv_update, conv_reset):
    """The update and the reset gate read the same [x, state] operand: their two 3x3 convolutions run as ONE with the output
    channels concatenated (exact).  The merged bf16 weight is cut from the two parameters' bf16 shadows once per optimizer
    step (all time steps of a GRU share it), not once per cell."""
    wu, wr = conv_update.weight, conv_reset.weight
    key = (id(wu), id(wr))
    stamp = (ops.weight_stamp(wu), ops.weight_stamp(wr), wu.data_ptr(), wr.data_ptr(),
             conv_update.bias._version, conv_reset.bias._version)
    ent = _GATE_WEIGHTS.get(key)
    # (id() values are recycled: an entry only counts while it still points at THESE live parameters)
    if ent is None or ent[0] != stamp or ent[3]() is not wu or ent[4]() is not wr:
        wb = torch.cat([ops._bf16_weights(wu)[0], ops._bf16_weights(wr)[0]], dim=0).contiguous(memory_format=torch.channels_last)
        bias = torch.cat([conv_update.bias.detach().float(), conv_reset.bias.detach().float()])
        if ent is None or ent[3]() is not wu:
            weakref.finalize(wu, _GATE_WEIGHTS.pop, key, None)
        ent = [stamp, wb, bias, weakref.ref(wu), weakref.ref(wr), None]
        _GATE_WEIGHTS[key] = ent
    return ent[1], ent[2]


def _merged_gate_weights_flipped(wu, wr, wb):
    """The merged gate weight with its taps flipped and its channel dimensions swapped (what the data gradient convolves dy
    with), once per optimizer step like the merged weight itself; cut on the spot when the cache has moved on."""
    ent = _GATE_WEIGHTS.get((id(wu), id(wr)))
    if ent is None or ent[1] is not wb:
        return wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
    if ent[5] is None:
        ent[5] = wb.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
    return ent[5]


def gru_cell_supported(x, state, conv_update, conv_reset, conv_state_tilde):
    """bf16 (or autocast) GPU tensors, 3x3 / stride 1 / padding 1 gate convolutions with a bias, channel counts multiples of 8."""
    bf16_autocast = torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
    if not (x.is_cuda and x.dim() == 4 and (bf16_autocast or (x.dtype == torch.bfloat16 and not torch.is_autocast_enabled()))):
        return False
    for conv in (conv_update, conv_reset, conv_state_tilde):
        if not (tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
                and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None
                and conv.weight.dtype == torch.float32):
            return False
    c, cx = state.shape[1], x.shape[1]
    return (cx % 8 == 0 and c % 8 == 0 and conv_update.in_channels == cx + c and conv_update.out_channels == c
            and conv_reset.out_channels == c and conv_state_tilde.out_channels == c
            and ops.conv2d_supported(x, conv_update.weight, 1))


def _gru_dims(xs, cx, c, bias_init):
    n, _, h, w = xs.shape
    return _lib.GruDims(n * h * w, cx, c, _lib.DTYPE_BF16, float(bias_init))


class _GruCell(torch.autograd.Function):
    """x (N,Cx,H,W), state (N,C,H,W) bf16 -> new state (N,C,H,W) bf16; all convolutions and the gate arithmetic inside, the
    cell's only saved tensors are [x | state], the gate pre-activations, [x | (1 - r) state] and the proposal (bf16)."""

    @staticmethod
    def forward(ctx, x, state, wu, bu, wr, br, wt, bt, mods, bias_init):
        conv_update, conv_reset, conv_tilde = mods
        ops._need_gpu(x, state)
        n, cx, h, w = x.shape
        c = state.shape[1]
        cl = torch.channels_last
        xs = torch.empty((n, cx + c, h, w), dtype=torch.bfloat16, device=x.device, memory_format=cl)
        torch.cat([x, state], dim=1, out=xs)
        wg_b, bg = _merged_gate_weights(conv_update, conv_reset)
        gates = ops._conv2d_launch(xs, wg_b, bg, 1, (1, 1), (1, 1), torch.bfloat16)
        dims = _gru_dims(xs, cx, c, bias_init)
        lib = _lib.lib()
        xs2 = torch.empty_like(xs, memory_format=cl)
        check(lib.stp3_gru_reset_cat_fwd(ctypes.byref(dims), ops._ptr(xs), ops._ptr(gates), ops._ptr(xs2), ops._stream()),
              'stp3_gru_reset_cat_fwd')
        wt_b = ops._bf16_weights(wt)[0]
        tilde = ops._conv2d_launch(xs2, wt_b, ops._f32(bt), 1, (1, 1), (1, 1), torch.bfloat16)
        out = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=x.device, memory_format=cl)
        check(lib.stp3_gru_output_fwd(ctypes.byref(dims), ops._ptr(gates), ops._ptr(xs), ops._ptr(tilde), ops._ptr(out),
                                      ops._stream()), 'stp3_gru_output_fwd')
        ctx.save_for_backward(xs, gates, xs2, tilde, wg_b, wt_b)
        ctx.params = (wu, wr, wt)
        ctx.stamps = tuple(ops.weight_stamp(p) for p in (wu, wr, wt))
        ctx.conf = (cx, c, float(bias_init), bu.dtype, br.dtype, bt.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        xs, gates, xs2, tilde, wg_b, wt_b = ctx.saved_tensors
        wu, wr, wt = ctx.params
        for p, st in zip(ctx.params, ctx.stamps):
            ops.check_weight_stamp(p, st, 'GRU cell backward')
        cx, c, bias_init, bu_dt, br_dt, bt_dt = ctx.conf
        cl = torch.channels_last
        n, _, h, w = xs.shape
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        dout, ld = ops._rows_view(dout)
        dims = _gru_dims(xs, cx, c, bias_init)
        lib = _lib.lib()
        dtilde = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=xs.device, memory_format=cl)
        dgates = torch.empty((n, 2 * c, h, w), dtype=torch.bfloat16, device=xs.device, memory_format=cl)
        check(lib.stp3_gru_output_bwd(ctypes.byref(dims), ops._ptr(dout), ld, ops._ptr(gates), ops._ptr(xs), ops._ptr(tilde),
                                      ops._ptr(dtilde), ops._ptr(dgates), ops._stream()), 'stp3_gru_output_bwd')
        pad, dil = (1, 1), (1, 1)
        need = ctx.needs_input_grad
        dwt = ops._conv2d_wgrad(dtilde, xs2, tuple(wt_b.shape), 1, pad, dil, leaf=wt) if need[6] else None
        dbt = ops.channel_sums(dtilde).to(bt_dt) if need[7] else None
        dxs2 = ops.conv2d_data_grad(dtilde, wt_b, wt, xs2.shape, 1, pad, dil)
        acc = torch.empty_like(xs, memory_format=cl)
        check(lib.stp3_gru_reset_cat_bwd(ctypes.byref(dims), ops._ptr(dout), ld, ops._ptr(gates), ops._ptr(xs),
                                         ops._ptr(dxs2), ops._ptr(acc), ops._ptr(dgates), ops._stream()),
              'stp3_gru_reset_cat_bwd')
        dwu = dwr = dbu = dbr = None
        if need[2] or need[4]:
            dwg = ops._conv2d_wgrad(dgates, xs, tuple(wg_b.shape), 1, pad, dil)
            dwu, dwr = dwg[:c], dwg[c:]
        if need[3] or need[5]:
            dbg = ops.channel_sums(dgates)
            dbu, dbr = dbg[:c].to(bu_dt), dbg[c:].to(br_dt)
        dx = dstate = None
        if need[0] or need[1]:
            # stride 1, padding 1: the data gradient is the 3x3 convolution of dgates with the flipped, channel-swapped weight
            dxs = ops._conv2d_launch(dgates, _merged_gate_weights_flipped(wu, wr, wg_b), None, 1, pad, dil, torch.bfloat16)
            total = torch.empty_like(xs, memory_format=cl)
            arr = (ctypes.c_void_p * 2)(acc.data_ptr(), dxs.data_ptr())
            check(lib.stp3_sum_n(2, total.numel(), _lib.DTYPE_BF16, arr, total.data_ptr(), ops._stream_handle()), 'stp3_sum_n')
            dx, dstate = total[:, :cx], total[:, cx:]
        return dx, dstate, dwu, dbu, dwr, dbr, dwt, dbt, None, None


_GRU_APPLY = ops._fast_apply(_GruCell)


def gru_cell(x, state, conv_update, conv_reset, conv_state_tilde, bias_init=0.0):
    """One convolutional GRU step on the kernels (``gru_cell_supported``): computed in bf16, the new state in ``state``'s type."""
    bf = torch.bfloat16
    state_dtype = state.dtype
    x = x if x.dtype == bf else x.to(bf)
    state = state if state.dtype == bf else state.to(bf)
    new_state = _GRU_APPLY(x, state, conv_update.weight, conv_update.bias, conv_reset.weight, conv_reset.bias,
                           conv_state_tilde.weight, conv_state_tilde.bias, (conv_update, conv_reset, conv_state_tilde),
                           float(bias_init))
    # the state leaves in the type it came in (what the torch statement of the cell, layers/temporal._gru_cell, returns)
    return new_state if state_dtype == bf else new_state.to(state_dtype)
