"""Fused operators of the camera trunk and the BEV networks on top of the C ABI:

``conv_bn_act``   convolution -> BatchNorm -> activation (-> + skip) as one autograd operator.  The epilogue of
                  ``stp3_conv2d_fwd`` (csrc/stp3_conv.hip) produces the BatchNorm statistics, so the forward makes ONE
                  pass over the convolution output less than ``ops.conv2d`` + ``ops.bn_act``:

                      y = act(BN(conv(x, w) + b) [+ res]) [+ res]

                  Replaces the same reference chains as those two operators (stp3/layers/convolutions.py:183-280,
                  stp3/layers/temporal.py:252-325, stp3/models/decoder.py:22-140, MBConv 1x1 -> BN -> swish).
                  Training mode with bf16 activations only; evaluation and float32 runs take the separate operators.
``se_block``      squeeze-and-excitation (EfficientNet MBConv): pooled mean, the two 1x1 layers and the gate.
"""
import ctypes
import os

import torch

from . import _lib, ops
from ._lib import check

_WS = {}
# the gate's two fully-connected layers through stp3_se_mlp_fwd / _bwd (one launch forward, one backward)
_SE_MLP = True


def _workspace(nbytes, device):
    key = ops._ws_key(device)                 # per device and stream (ops._ws_key)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 32 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def conv2d_v2(x, wb, bias, stride, pad, dil, sums_ptr=None):
    """Raw launch: x (N,Cin,H,W) bf16 channels-last view, wb (Cout,Cin,KH,KW) bf16 channels-last -> y bf16.
    ``sums_ptr``: device address of a float32 [2][Cout] buffer that receives the BatchNorm statistics of y."""
    return ops._conv2d_launch(x, wb, bias, stride, pad, dil, torch.bfloat16, sums_ptr=sums_ptr)


class _ConvBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, cbias, gamma, beta, res, running_mean, running_var, momentum, eps, act, res_mode,
                stride, pad, dil, group, oscale=None, out_slot=None, res_carrier=None, skip_carrier=None):
        return ops.drive_exchange(_ConvBnAct.forward_steps(ctx, x, weight, cbias, gamma, beta, res, running_mean,
                                                           running_var, momentum, eps, act, res_mode, stride, pad, dil,
                                                           group, oscale, out_slot, res_carrier, skip_carrier), group)

    @staticmethod
    def backward(ctx, dy):
        return ops.drive_exchange(_ConvBnAct.backward_steps(ctx, dy), ctx.cfg[3])

    @staticmethod
    def forward_steps(ctx, x, weight, cbias, gamma, beta, res, running_mean, running_var, momentum, eps, act, res_mode,
                      stride, pad, dil, group, oscale=None, out_slot=None, res_carrier=None, skip_carrier=None):
        """``res_carrier`` / ``skip_carrier`` (ops.SkipCarrier): this operator ADDS a block's skip (res, RES_AFTER_ACT) and hands
        the skip's gradient to the carrier instead of returning it / this operator CONSUMES the block input and adds the
        carrier's gradient to its data gradient in the kernel that writes it.
        ``forward`` as a generator (ops.drive_exchange): yields the [2C] statistics when they need the other replicas.
        ``out_slot`` = (buffer, first channel): the result is written into that channel slice of a wider channels-last
        tensor (a concatenation that is never copied: ``join_slices``) instead of a tensor of its own."""
        ops._need_gpu(x, weight)
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        wb, _ = ops._bf16_weights(weight)
        cout = wb.shape[0]
        dev = x.device
        stat = torch.empty(4 * cout, dtype=torch.float32, device=dev)       # sum | sum of squares | mean | invstd
        base = stat.data_ptr()
        yc = conv2d_v2(x, wb, ops._f32(cbias), stride, pad, dil, sums_ptr=base)
        n, _, ho, wo = yc.shape
        count = float(n * ho * wo)
        world, exchange = ops.replicas(group)
        if exchange:
            yield stat[:2 * cout]
            count *= world
        ldr = cout
        if res is not None:
            res, ldr = ops._rows_view(res if res.dtype == torch.bfloat16 else res.to(torch.bfloat16))
        else:
            res_mode = ops.RES_NONE
        ldo = cout
        if out_slot is None:
            out = torch.empty_like(yc)
        else:
            out, ldo = slot_view(out_slot, yc)
        osc = ops._f32(oscale)
        dims = _lib.BnDims(n, ho * wo, cout, cout, ldo, ldr, _lib.DTYPE_BF16, act, res_mode, 0, int(osc is not None))
        g32, b32 = ops._f32(gamma), ops._f32(beta)
        check(_lib.lib().stp3_bn_apply_fwd(ctypes.byref(dims), yc.data_ptr(), None, ops._opt_ptr(res), ops._opt_ptr(osc), base, count,
                                           ops._opt_ptr(g32), ops._opt_ptr(b32), eps, momentum, ops._opt_ptr(running_mean),
                                           ops._opt_ptr(running_var), base + 8 * cout, base + 12 * cout, out.data_ptr(),
                                           ops._stream_handle()), 'stp3_bn_apply_fwd')
        ctx.save_for_backward(x, wb, yc, res if res_mode == ops.RES_BEFORE_ACT else None, g32, b32, stat, osc)
        if ldo != cout:                         # the backward passes read dy with ITS row stride (set there)
            dims = _lib.BnDims(n, ho * wo, cout, cout, cout, ldr, _lib.DTYPE_BF16, act, res_mode, 0, int(osc is not None))
        ctx.cfg = (dims, count, exchange, group, stride, pad, dil, cbias is not None)
        ctx.carriers = (res_carrier if (res is not None and res_mode == ops.RES_AFTER_ACT) else None, skip_carrier)
        ctx.weight_ref = ops.note_weight_use(weight)
        ctx.weight_stamp = ops.weight_stamp(weight)
        ctx.dtypes = (weight.dtype, None if cbias is None else cbias.dtype, None if gamma is None else gamma.dtype,
                      None if beta is None else beta.dtype, None if res is None else res.dtype)
        return out

    @staticmethod
    def backward_steps(ctx, dy):
        x, wb, yc, res, g32, b32, stat, osc = ctx.saved_tensors
        ops.check_weight_stamp(ctx.weight_ref, ctx.weight_stamp, 'conv_bn_act backward')
        dims, count, exchange, group, stride, pad, dil, has_cbias = ctx.cfg
        wdt, cbdt, gdt, bdt, rdt = ctx.dtypes
        res_carrier, skip_carrier = ctx.carriers
        n, rows, c = dims.N, dims.rows, dims.C
        dev = x.device
        lib = _lib.lib()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy, ldy = ops._rows_view(dy)
        if ldy != c:
            if ldy % 8 == 0 and dy.data_ptr() % 16 == 0:
                # a channel slice of a wider channels-last tensor (the gradient of a concatenation): read in place with
                # its own row stride instead of being copied dense first
                dims = _lib.BnDims(dims.N, dims.rows, dims.C, dims.ldx, ldy, dims.ldr, dims.dtype, dims.act, dims.res_mode, 0,
                                   dims.has_oscale)
            else:
                dy = dy.contiguous(memory_format=torch.channels_last)
        ws, ws_bytes = ops._bn_workspace(n, c, dev)
        sumbuf = torch.empty((n + 1) * 3 * c, dtype=torch.float32, device=dev)
        sums_off = n * 3 * c
        mean_p, invstd_p = stat.data_ptr() + 8 * c, stat.data_ptr() + 12 * c
        dconv = torch.empty_like(yc)                                   # gradient at the convolution output
        dres = None
        if dims.res_mode == ops.RES_BEFORE_ACT and ctx.needs_input_grad[5]:
            dres = torch.empty_like(yc)
            if dims.ldr != c:
                res = res.contiguous(memory_format=torch.channels_last)
                dims = _lib.BnDims(dims.N, dims.rows, dims.C, dims.ldx, dims.ldy, c, dims.dtype, dims.act, dims.res_mode, 0,
                                   dims.has_oscale)
        if dims.res_mode == ops.RES_AFTER_ACT and dims.ldy != c:
            dy = dy.contiguous(memory_format=torch.channels_last)       # dres = dy is handed back: dense
            dims = _lib.BnDims(dims.N, dims.rows, dims.C, dims.ldx, c, dims.ldr, dims.dtype, dims.act, dims.res_mode, 0,
                               dims.has_oscale)
        stream = ops._stream_handle()
        check(lib.stp3_bn_bwd_reduce(ctypes.byref(dims), dy.data_ptr(), yc.data_ptr(), None, ops._opt_ptr(res), ops._opt_ptr(osc), mean_p,
                                     invstd_p, ops._opt_ptr(g32), ops._opt_ptr(b32), ws.data_ptr(), ws_bytes,
                                     sumbuf.data_ptr(), sumbuf.data_ptr() + 4 * sums_off, stream), 'stp3_bn_bwd_reduce')
        lsums = sumbuf[sums_off:].view(3, c)
        gsums = lsums
        if exchange:
            gsums = lsums.clone()
            yield gsums
        check(lib.stp3_bn_apply_bwd(ctypes.byref(dims), dy.data_ptr(), yc.data_ptr(), None, ops._opt_ptr(res), ops._opt_ptr(osc), mean_p,
                                    invstd_p, ops._opt_ptr(g32), ops._opt_ptr(b32), gsums.data_ptr(), count,
                                    dconv.data_ptr(), ops._opt_ptr(dres), stream), 'stp3_bn_apply_bwd')
        dgamma = lsums[1].to(gdt) if gdt is not None and ctx.needs_input_grad[3] else None
        dbeta = lsums[0].to(bdt) if bdt is not None and ctx.needs_input_grad[4] else None
        if dims.res_mode == ops.RES_AFTER_ACT and ctx.needs_input_grad[5]:
            dres = dy
        if dres is not None and rdt is not None and dres.dtype != rdt:
            dres = dres.to(rdt)
        if res_carrier is not None and dres is not None:
            res_carrier.grad, dres = dres, None          # (the block's first operator adds it to its data gradient)
        skip_grad = None if skip_carrier is None else skip_carrier.take()
        # ---- convolution backward (same routes as ops._Conv2dMfma.backward) ------------------------------------
        cout, cin, kh, kw = wb.shape
        dx = dw = dcb = None
        bpad = (dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1])
        need_dx = ctx.needs_input_grad[0]
        hip_dx = need_dx and cout % 8 == 0 and bpad[0] >= 0 and bpad[1] >= 0
        if hip_dx:
            dx = ops.conv2d_data_grad(dconv, wb, ctx.weight_ref, x.shape, stride, pad, dil, add=skip_grad)
            skip_grad = None
        need_dw = ctx.needs_input_grad[1]
        need_db = has_cbias and ctx.needs_input_grad[2]
        hip_dw = need_dw and cin % 8 == 0 and cout % 8 == 0
        if hip_dw:
            dw = ops._conv2d_wgrad(dconv, x, (cout, cin, kh, kw), stride, pad, dil, leaf=ctx.weight_ref).to(wdt)
            if need_db:
                dcb = ops.channel_sums(dconv).to(cbdt)
        mask = [need_dx and not hip_dx, need_dw and not hip_dw, need_db and not hip_dw]
        if any(mask):
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            gx, gw, gb = torch.ops.aten.convolution_backward(dconv, xd, wb, [cout] if has_cbias else None, [stride, stride],
                                                             list(pad), list(dil), False, [0, 0], 1, mask)
            if mask[0]:
                dx = gx
            if mask[1]:
                dw = gw.to(wdt)
            if mask[2]:
                dcb = gb.to(cbdt)
        if skip_grad is not None:
            dx = skip_grad if dx is None else dx + skip_grad.to(dx.dtype)
        return (dx, dw, dcb, dgamma, dbeta, dres) + (None,) * 14


# the expand convolution's data gradient computed inside the BatchNorm-backward apply pass (stp3_conv2d_bn_bwd_apply_dx)
EXPAND_DGRAD_IN_APPLY = True


class _PointwiseBnAct(torch.autograd.Function):
    """1x1 convolution -> BatchNorm -> activation WITHOUT the convolution output in memory: for the expand convolutions
    of the MBConv blocks (24..160 -> 144..960 channels), whose output is 6x their input.  Forward: the convolution runs
    twice (stp3_conv2d_fwd_stats: BatchNorm statistics only; stp3_conv2d_fwd_bnact: the same tiles again, act(BN(.))
    written) -- the expanded pre-activation tensor E0 is neither written nor read back.  Backward: the two passes of the
    BatchNorm backward recompute the E0 tiles they need from the block input (stp3_conv2d_bn_bwd_reduce / _bwd_apply) and
    read only the incoming gradient.  Passes over an expanded tensor: 1 forward and 3 backward where ``_ConvBnAct`` makes 3
    and 5 (the data and weight gradient of the convolution follow unchanged); E0 is not kept for the backward pass at all.
    Same values: every kernel rounds the accumulators to bf16 before it uses them, as the stored tensor was."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, momentum, eps, act, group, skip_carrier=None):
        return ops.drive_exchange(_PointwiseBnAct.forward_steps(ctx, x, weight, gamma, beta, running_mean, running_var,
                                                                momentum, eps, act, group, skip_carrier), group)

    @staticmethod
    def backward(ctx, dz):
        return ops.drive_exchange(_PointwiseBnAct.backward_steps(ctx, dz), ctx.cfg[3])

    @staticmethod
    def forward_steps(ctx, x, weight, gamma, beta, running_mean, running_var, momentum, eps, act, group, skip_carrier=None):
        ops._need_gpu(x, weight)
        ctx.skip_carrier = skip_carrier           # (ops.SkipCarrier: the block's skip gradient joins the data gradient)
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        wb, _ = ops._bf16_weights(weight)
        cout, cin = wb.shape[:2]
        n, _, h, w = x.shape
        x, ldx = ops._rows_view(x)
        dev = x.device
        lib = _lib.lib()
        dims = _lib.ConvDims(n, h, w, cin, h, w, cout, 1, 1, 1, 0, 0, 1, 1, ldx, cout, _lib.DTYPE_BF16, 0)
        need = ctypes.c_size_t()
        check(lib.stp3_conv2d_fwd_workspace(ctypes.byref(dims), ctypes.byref(need)), 'stp3_conv2d_fwd_workspace')
        ws = _workspace(need.value, dev)
        stat = torch.empty(6 * cout, dtype=torch.float32, device=dev)       # sum | sum of squares | scale | shift | mean | invstd
        stream = ops._stream_handle()
        check(lib.stp3_conv2d_fwd_stats(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), stat.data_ptr(), ws.data_ptr(),
                                        need.value, stream), 'stp3_conv2d_fwd_stats')
        count = float(n * h * w)
        world, exchange = ops.replicas(group)
        if exchange:
            yield stat[:2 * cout]
            count *= world
        g32, b32 = ops._f32(gamma), ops._f32(beta)
        coef = stat[2 * cout:]
        check(lib.stp3_bn_finalize(stat.data_ptr(), cout, count, ops._opt_ptr(g32), ops._opt_ptr(b32), eps, momentum,
                                   ops._opt_ptr(running_mean), ops._opt_ptr(running_var), coef.data_ptr(), stream),
              'stp3_bn_finalize')
        y = torch.empty((n, cout, h, w), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
        check(lib.stp3_conv2d_fwd_bnact(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), coef.data_ptr(), int(act), y.data_ptr(),
                                        stream), 'stp3_conv2d_fwd_bnact')
        ctx.save_for_backward(x, wb, coef)
        ctx.cfg = (dims, count, exchange, group, int(act), need.value)
        ctx.weight_ref = ops.note_weight_use(weight)
        ctx.weight_stamp = ops.weight_stamp(weight)
        ctx.dtypes = (weight.dtype, None if gamma is None else gamma.dtype, None if beta is None else beta.dtype)
        return y

    @staticmethod
    def backward_steps(ctx, dz):
        x, wb, coef = ctx.saved_tensors
        ops.check_weight_stamp(ctx.weight_ref, ctx.weight_stamp, 'pointwise_bn_act backward')
        dims, count, exchange, group, act, ws_bytes = ctx.cfg
        wdt, gdt, bdt = ctx.dtypes
        cout, cin = wb.shape[:2]
        dev = x.device
        lib = _lib.lib()
        if dz.dtype != torch.bfloat16:
            dz = dz.to(torch.bfloat16)
        dz, ldz = ops._rows_view(dz)
        if ldz % 8 or dz.data_ptr() % 16:
            dz, ldz = dz.contiguous(memory_format=torch.channels_last), cout
        ws = _workspace(ws_bytes, dev)
        stream = ops._stream_handle()
        lsums = torch.empty(2, cout, dtype=torch.float32, device=dev)        # sum g (dbeta) | sum g * xhat (dgamma)
        check(lib.stp3_conv2d_bn_bwd_reduce(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), dz.data_ptr(), ldz, coef.data_ptr(),
                                            act, lsums.data_ptr(), ws.data_ptr(), ws_bytes, stream), 'stp3_conv2d_bn_bwd_reduce')
        gsums = lsums
        if exchange:
            gsums = lsums.clone()
            yield gsums
        dconv = torch.empty((dims.N, cout, dims.H, dims.W), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
        dx = dw = None
        skip_grad = None if ctx.skip_carrier is None else ctx.skip_carrier.take()
        applied = False
        if ctx.needs_input_grad[0] and EXPAND_DGRAD_IN_APPLY and cout in (144, 192) and cin <= 32:
            # the data gradient out of the same pass: the kernel holds the gradient tile in LDS when it stores it (whole-row
            # streaming kernel only: 144 / 192 channels from <= 32 -- the launcher's own test, repeated here so that no buffer
            # is made for nothing; a shape it still turns down answers STP3_EUNSUP and takes the two calls)
            dx = torch.empty((dims.N, cin, dims.H, dims.W), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
            add, ldadd = (None, 0)
            if skip_grad is not None:
                add, ldadd = ops._rows_view(skip_grad if skip_grad.dtype == torch.bfloat16 else skip_grad.to(torch.bfloat16))
            rc = lib.stp3_conv2d_bn_bwd_apply_dx(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), dz.data_ptr(), ldz, coef.data_ptr(),
                                                 act, gsums.data_ptr(), count, dconv.data_ptr(), dx.data_ptr(), cin,
                                                 ops._opt_ptr(add), ldadd, stream)
            if rc == -10002:                                  # STP3_EUNSUP: not a shape of that kernel
                dx = None
            else:
                check(rc, 'stp3_conv2d_bn_bwd_apply_dx')
                applied, skip_grad = True, None
        if not applied:
            check(lib.stp3_conv2d_bn_bwd_apply(ctypes.byref(dims), x.data_ptr(), wb.data_ptr(), dz.data_ptr(), ldz, coef.data_ptr(),
                                               act, gsums.data_ptr(), count, dconv.data_ptr(), stream), 'stp3_conv2d_bn_bwd_apply')
        dgamma = lsums[1].to(gdt) if gdt is not None and ctx.needs_input_grad[2] else None
        dbeta = lsums[0].to(bdt) if bdt is not None and ctx.needs_input_grad[3] else None
        if ctx.needs_input_grad[0] and dx is None:
            dx = ops.conv2d_data_grad(dconv, wb, ctx.weight_ref, x.shape, 1, (0, 0), (1, 1), add=skip_grad)
        elif dx is None and skip_grad is not None:
            dx = skip_grad
        if ctx.needs_input_grad[1]:
            dw = ops._conv2d_wgrad(dconv, x, (cout, cin, 1, 1), 1, (0, 0), (1, 1), leaf=ctx.weight_ref).to(wdt)
        return (dx, dw, dgamma, dbeta) + (None,) * 7


def pointwise_bn_act_supported(x, conv, bn):
    """``_PointwiseBnAct``: training-mode BatchNorm with running statistics behind a bias-free 1x1 / stride-1 convolution
    of bf16 GPU activations, channel counts in whole 16-byte pieces, within the statistics epilogue's tile bound."""
    if not (x.is_cuda and x.dim() == 4 and bn.training and bn.track_running_stats and conv.bias is None):
        return False
    if tuple(conv.kernel_size) != (1, 1) or tuple(conv.stride) != (1, 1) or conv.groups != 1:
        return False
    cout, cin = conv.weight.shape[:2]
    n, _, h, w = x.shape
    return cin % 8 == 0 and cout % 8 == 0 and (n * h * w + 127) // 128 <= ops._CONV_MAX_STAT_TILES and \
        ops.conv2d_supported(x, conv.weight, 1)


def pointwise_bn_act_pays(x, conv):
    """Whether the recomputing route is the faster one: where the library runs the four passes on its streaming kernels
    (stp3_conv.hip: pointwise_rows_kernel / pointwise_direct_kernel -- contraction <= 128, >= 16384 pixels).  On the tiled
    kernel a recomputing pass costs as much as a storing one (160 -> 960 @14x30x72: BatchNorm-backward passes of 75 + 51 us
    against ~20 + 25 us on the stored 58-MB tensor)."""
    n, cin, h, w = x.shape
    return cin <= 128 and conv.weight.shape[0] >= 64 and n * h * w >= 16384


def pointwise_bn_act(x, conv, bn, act, group=None, skip_carrier=None):
    """act(bn(conv(x))) for a 1x1 convolution through ``_PointwiseBnAct`` (see there)."""
    if bn.num_batches_tracked is not None:
        ops.bump_batch_counter(bn)
    return _PointwiseBnAct.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, ops.bn_momentum(bn),
                                 float(bn.eps), int(act), group, skip_carrier)


slot_view = ops.slot_view


class _JoinSlices(torch.autograd.Function):
    """The concatenation along the channels of results that were WRITTEN into the channel slices of one buffer
    (``out_slot``): returns the buffer -- no copy forward; backward hands every member its channel slice of the gradient
    (a view: the BatchNorm backward passes read it in place)."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        c0 = 0
        for p in parts:
            if p.data_ptr() != buf.data_ptr() + c0 * buf.element_size() or p.shape[1] + c0 > buf.shape[1]:
                raise _lib.Stp3HipError('join_slices: a part does not lie in its slot')
            c0 += p.shape[1]
        if c0 != buf.shape[1]:
            raise _lib.Stp3HipError('join_slices: the parts do not fill the buffer')
        ctx.sizes = [p.shape[1] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g.split(ctx.sizes, dim=1))


class _CopyIntoSlot(torch.autograd.Function):
    """A tensor that some other operator produced, copied into its channel slice of a concatenation's buffer (the members
    that CAN write their slice directly do: ``out_slot``); the gradient of the slice goes back as it is."""

    @staticmethod
    def forward(ctx, x, holder, c0):
        view, _ = slot_view((holder[0], c0), x)
        view.copy_(x)
        return view

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def copy_into_slot(x, out_slot):
    return _CopyIntoSlot.apply(x, (out_slot[0],), out_slot[1])


def join_slices(buf, parts):
    return _JoinSlices.apply((buf,), *parts)


def conv_bn_act(x, weight, cbias, bn, act=ops.ACT_NONE, res=None, res_mode=ops.RES_NONE, stride=1, padding=0, dilation=1,
                group=None, oscale=None, out_slot=None, res_carrier=None, skip_carrier=None):
    """Training-mode conv -> BatchNorm -> activation (-> + skip) through the fused kernels (GPU, bf16)."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        ops.bump_batch_counter(bn)
    s = ops._pair(stride)
    return _ConvBnAct.apply(x, weight, cbias, bn.weight, bn.bias, res, bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None,
                            ops.bn_momentum(bn), float(bn.eps), int(act), int(res_mode),
                            s[0], ops._pair(padding), ops._pair(dilation), group, oscale, out_slot, res_carrier, skip_carrier)


class _SubContext:
    """What one member operator of an ``_ExchangeGroup`` sees in place of the autograd context."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


_MEMBER_OPS = {'bn_act': ops._BnAct, 'conv_bn_act': _ConvBnAct}


class _ExchangeGroup(torch.autograd.Function):
    """SIBLING BatchNorm operators -- parallel branches that read the same tensor: the five ASPP branches, the
    pointwise convolutions and the pooled descriptor at the head of a temporal block, the decoder heads, a ResNet block's
    down-sampling skip and first convolution -- as ONE autograd node, so that their
    cross-replica statistics travel in one all-reduce per pass instead of one per layer (forward [2C] sums, backward
    [3C] sums; 52 of the 258 exchanges of a BASELINE configs[2] step at N > 1).  The members are the ordinary operators run as generators
    (``forward_steps`` / ``backward_steps``): same kernels, same arithmetic, same order per member."""

    @staticmethod
    def forward(ctx, specs, group, *flat):
        subs, gens, off = [], [], 0
        for kind, n_args in specs:
            sub = _SubContext(ctx.needs_input_grad[2 + off:2 + off + n_args])
            gens.append(_MEMBER_OPS[kind].forward_steps(sub, *flat[off:off + n_args]))
            subs.append(sub)
            off += n_args
        outs = ops.drive_exchange_group(gens, group)
        saved, index = [], []
        for sub in subs:
            index.append((len(saved), len(sub.saved_tensors)))
            saved.extend(sub.saved_tensors)
            sub.saved_tensors = ()
        ctx.save_for_backward(*saved)
        ctx.members = (subs, index, specs, group)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        subs, index, specs, group = ctx.members
        saved = ctx.saved_tensors
        gens = []
        for sub, (first, count), (kind, _), dy in zip(subs, index, specs, dys):
            if dy is None:
                raise RuntimeError('an output of a BatchNorm exchange group received no gradient')
            sub.saved_tensors = saved[first:first + count]
            gens.append(_MEMBER_OPS[kind].backward_steps(sub, dy))
        grads = ops.drive_exchange_group(gens, group)
        flat = []
        for (kind, n_args), g in zip(specs, grads):
            assert len(g) >= n_args and all(v is None for v in g[n_args:]), (kind, len(g), n_args)
            flat.extend(g[:n_args])                      # (trailing optional arguments the member was not given)
        return (None, None) + tuple(flat)


def exchange_group(members, group=None):
    """``members``: [('bn_act' | 'conv_bn_act', positional arguments of that operator's autograd function)] ->
    their outputs, computed with one statistics exchange per pass for all of them."""
    specs = tuple((kind, len(args)) for kind, args in members)
    flat = [a for _, args in members for a in args]
    return list(_ExchangeGroup.apply(specs, group, *flat))


# ----------------------------------------------------------------------------------------------
# squeeze-and-excitation block as one operator (EXPERIMENTAL, STP3_FUSED_SE=1)
# ----------------------------------------------------------------------------------------------
def _se_dims(x):
    n, c, h, w = x.shape
    x, ld = ops._rows_view(x)
    if x.dtype == torch.bfloat16:
        dt = _lib.DTYPE_BF16
    elif x.dtype == torch.float32:
        dt = _lib.DTYPE_F32
    else:
        raise _lib.Stp3HipError(f'se_block supports float32 / bfloat16, got {x.dtype}')
    return x, _lib.SeDims(n, h * w, c, ld, dt)


def _se_pool(x, dims, dy=None):
    lib = _lib.lib()
    nbytes = ctypes.c_size_t()
    check(lib.stp3_se_workspace_bytes(ctypes.byref(dims), ctypes.byref(nbytes)), 'stp3_se_workspace_bytes')
    ws = _workspace(nbytes.value, x.device)
    out = torch.empty(dims.N, dims.C, dtype=torch.float32, device=x.device)
    check(lib.stp3_se_pool(ctypes.byref(dims), x.data_ptr(), ops._opt_ptr(dy), ws.data_ptr(), nbytes.value, out.data_ptr(),
                           ops._stream_handle()), 'stp3_se_pool')
    return out


def _se_scale(x, dims, gate, add=None):
    y = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
    check(_lib.lib().stp3_se_scale(ctypes.byref(dims), x.data_ptr(), gate.data_ptr(), ops._opt_ptr(add), y.data_ptr(),
                                   ops._stream_handle()), 'stp3_se_scale')
    return y


class _SeBlock(torch.autograd.Function):
    """y = x * sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2): pooling and gating through stp3_se_pool / stp3_se_scale,
    the two tiny fully-connected layers in float32 torch ops on (N, C) tensors."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        ops._need_gpu(x)
        x, dims = _se_dims(x)
        hw = float(dims.rows)
        if _SE_MLP:                                                           # gate MLP as one launch
            pooled_sum = _se_pool(x, dims)
            w1f, w2f = w1.detach().flatten(1).float().contiguous(), w2.detach().flatten(1).float().contiguous()
            md = _lib.SeMlpDims(dims.N, dims.C, w1f.shape[0], 1.0 / hw)
            z1 = torch.empty(dims.N, md.S, dtype=torch.float32, device=x.device)
            gate = torch.empty(dims.N, dims.C, dtype=torch.float32, device=x.device)
            check(_lib.lib().stp3_se_mlp_fwd(ctypes.byref(md), pooled_sum.data_ptr(), w1f.data_ptr(),
                                             ops._f32(b1).data_ptr(), w2f.data_ptr(), ops._f32(b2).data_ptr(),
                                             z1.data_ptr(), gate.data_ptr(), ops._stream_handle()), 'stp3_se_mlp_fwd')
            y = _se_scale(x, dims, gate)
            ctx.save_for_backward(x, gate, pooled_sum, z1, w1f, w2f)
            ctx.dims, ctx.mlp_dims = dims, md
            ctx.meta = (w1.shape, w2.shape, w1.dtype, b1.dtype, w2.dtype, b2.dtype)
            return y
        pooled = _se_pool(x, dims) / hw                                       # (N, C)
        w1f, w2f = w1.detach().flatten(1).float(), w2.detach().flatten(1).float()
        with torch.autocast('cuda', enabled=False):                           # the kernels read float32 gates
            z1 = torch.addmm(b1.detach().float(), pooled, w1f.t())           # (N, S)
            h = torch.nn.functional.silu(z1)
            gate = torch.sigmoid(torch.addmm(b2.detach().float(), h, w2f.t()))   # (N, C)
        y = _se_scale(x, dims, gate)
        ctx.save_for_backward(x, gate, pooled, z1, h, w1f, w2f)
        ctx.dims, ctx.mlp_dims = dims, None
        ctx.meta = (w1.shape, w2.shape, w1.dtype, b1.dtype, w2.dtype, b2.dtype)
        return y

    @staticmethod
    def _backward_mlp_kernels(ctx, dy):
        x, gate, pooled_sum, z1, w1f, w2f = ctx.saved_tensors
        dims, md = ctx.dims, ctx.mlp_dims
        w1s, w2s, w1d, b1d, w2d, b2d = ctx.meta
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dy.stride() != x.stride():
            dy = dy.contiguous(memory_format=torch.channels_last)
            if dy.stride() != x.stride():                                     # x was a channel-sliced view
                x, dims = _se_dims(x.contiguous(memory_format=torch.channels_last))
        dgate = _se_pool(x, dims, dy)                                         # sum_hw dy * x
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dz2, dz1 = torch.empty(md.N, md.C, **f32), torch.empty(md.N, md.S, **f32)
        dpooled = torch.empty(md.N, md.C, **f32)
        dw1, db1 = torch.empty(md.S, md.C, **f32), torch.empty(md.S, **f32)
        dw2, db2 = torch.empty(md.C, md.S, **f32), torch.empty(md.C, **f32)
        check(_lib.lib().stp3_se_mlp_bwd(ctypes.byref(md), dgate.data_ptr(), gate.data_ptr(), pooled_sum.data_ptr(),
                                         z1.data_ptr(), w1f.data_ptr(), w2f.data_ptr(), dz2.data_ptr(), dz1.data_ptr(),
                                         dpooled.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                         db2.data_ptr(), ops._stream_handle()), 'stp3_se_mlp_bwd')
        dx = _se_scale(dy, dims, gate, dpooled)
        return dx, dw1.view(w1s).to(w1d), db1.to(b1d), dw2.view(w2s).to(w2d), db2.to(b2d)

    @staticmethod
    def backward(ctx, dy):
        if ctx.mlp_dims is not None:
            return _SeBlock._backward_mlp_kernels(ctx, dy)
        x, gate, pooled, z1, h, w1f, w2f = ctx.saved_tensors
        dims = ctx.dims
        w1s, w2s, w1d, b1d, w2d, b2d = ctx.meta
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dy.stride() != x.stride():
            dy = dy.contiguous(memory_format=torch.channels_last)
            if dy.stride() != x.stride():                                     # x was a channel-sliced view
                xd = x.contiguous(memory_format=torch.channels_last)
                x, dims = _se_dims(xd)
        dgate = _se_pool(x, dims, dy)                                         # sum_hw dy * x
        with torch.autocast('cuda', enabled=False):
            return _SeBlock._mlp_backward(dims, dy, dgate, gate, pooled, z1, h, w1f, w2f, ctx.meta)

    @staticmethod
    def _mlp_backward(dims, dy, dgate, gate, pooled, z1, h, w1f, w2f, meta):
        w1s, w2s, w1d, b1d, w2d, b2d = meta
        dz2 = dgate * gate * (1.0 - gate)
        dw2 = dz2.t().mm(h)
        db2 = dz2.sum(0)
        dh = dz2.mm(w2f)
        sg = torch.sigmoid(z1)
        dz1 = dh * sg * (1.0 + z1 * (1.0 - sg))
        dw1 = dz1.t().mm(pooled)
        db1 = dz1.sum(0)
        dpooled = dz1.mm(w1f) / float(dims.rows)
        dx = _se_scale(dy, dims, gate, dpooled.contiguous())
        return dx, dw1.view(w1s).to(w1d), db1.to(b1d), dw2.view(w2s).to(w2d), db2.to(b2d)


def se_block(x, se_reduce, se_expand):
    """Squeeze-and-excitation of an MBConv block; ``se_reduce`` / ``se_expand`` are its two 1x1 conv modules."""
    return _SeBlock.apply(x, se_reduce.weight, se_reduce.bias, se_expand.weight, se_expand.bias)


# ----------------------------------------------------------------------------------------------
# MBConv middle: depthwise conv -> BatchNorm -> swish -> squeeze-excite, as ONE operator
# ----------------------------------------------------------------------------------------------
# The BatchNorm-1 constants of the MBConv middle are made by the reduction that produces their sums
# (stp3_dwconv2d_fwd_stats_bn: one launch less per block, bit-identical).  The same idea one step further -- the gate kernels
# adding the squeeze's / the backward sums' partial rows themselves instead of waiting for mb_reduce_kernel -- was built and
# measured and LOSES 0.65 ms per step: those consumers are one workgroup per sample (profiles/r06e_ab_consumer_side_reductions.txt).
MERGE_SMALL_REDUCTIONS = True


class _DwBnSe(torch.autograd.Function):
    """A = swish(BN1(depthwise(x))) * gate,  gate = sigmoid(W2 swish(W1 mean_hw swish(BN1(.)) + b1) + b2)
    (efficientnet_pytorch MBConvBlock between the expand and the project convolution, driven by
    stp3/models/encoder.py:57-97), training mode, without ever writing S = swish(BN1(.)):

      forward   stp3_dwconv2d_fwd_stats (E2 + the BatchNorm statistics from its epilogue) -> stp3_bn_finalize ->
                stp3_se_pool_act (squeeze straight from E2) -> stp3_se_mlp_fwd -> stp3_mbconv_scale_act (A from E2)
      backward  stp3_mbconv_bwd_reduce (ONE pass over dA, E2 for the gate gradient and the BatchNorm reductions) ->
                stp3_se_mlp_bwd -> stp3_mbconv_bwd_coef -> stp3_mbconv_bwd_apply (dE2) -> depthwise data / weight gradients

    5 passes over the expanded tensor forward (8 as separate operators), 9 backward (13); saves x and E2 only (not S, A).
    With more than one rank the statistics and the two backward sums are all-reduced like ``ops.bn_act`` does."""

    @staticmethod
    def forward(ctx, x, dw_weight, stride, pad, gamma, beta, running_mean, running_var, momentum, eps, w1, b1, w2, b2, group):
        ops._need_gpu(x, dw_weight)
        lib = _lib.lib()
        c, _, k, _ = dw_weight.shape
        left, right, top, bottom = pad
        n, _, h, w = x.shape
        ho = (h + top + bottom - k) // stride + 1
        wo = (w + left + right - k) // stride + 1
        x = x.contiguous(memory_format=torch.channels_last)
        dev = x.device
        wt = ops._dw_weight_taps(dw_weight)                                              # [K*K][C] float32, cached per step
        e2 = torch.empty((n, c, ho, wo), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        dwd = ops._dw_dims(x, k, stride, top, left, ho, wo)
        sd = _lib.SeDims(n, ho * wo, c, c, dwd.dtype)
        need = ctypes.c_size_t()
        check(lib.stp3_dwconv2d_fwd_stats_workspace(ctypes.byref(dwd), ctypes.byref(need)), 'stp3_dwconv2d_fwd_stats_workspace')
        need2 = ctypes.c_size_t()
        check(lib.stp3_mbconv_workspace_bytes(ctypes.byref(sd), ctypes.byref(need2)), 'stp3_mbconv_workspace_bytes')
        ws_bytes = max(need.value, need2.value)
        ws = _workspace(ws_bytes, dev)
        stream = ops._stream_handle()
        stat = torch.empty(6 * c, dtype=torch.float32, device=dev)        # sum | sum of squares | scale | shift | mean | invstd
        count = float(n * ho * wo)
        world, exchange = ops.replicas(group)
        g32, b32 = ops._f32(gamma), ops._f32(beta)
        coef = stat[2 * c:]
        if exchange or not MERGE_SMALL_REDUCTIONS:
            check(lib.stp3_dwconv2d_fwd_stats(ctypes.byref(dwd), x.data_ptr(), wt.data_ptr(), e2.data_ptr(), stat.data_ptr(),
                                              ws.data_ptr(), ws_bytes, stream), 'stp3_dwconv2d_fwd_stats')
            if exchange:
                torch.distributed.all_reduce(stat[:2 * c], group=group)
                ops._EXCHANGES['batchnorm'] += 1
                count *= world
            check(lib.stp3_bn_finalize(stat.data_ptr(), c, count, ops._opt_ptr(g32), ops._opt_ptr(b32), eps, momentum,
                                       ops._opt_ptr(running_mean), ops._opt_ptr(running_var), coef.data_ptr(), stream),
                  'stp3_bn_finalize')
        else:
            # nothing happens between the statistics and their use: the last reduction finishes its channels itself
            check(lib.stp3_dwconv2d_fwd_stats_bn(ctypes.byref(dwd), x.data_ptr(), wt.data_ptr(), e2.data_ptr(), stat.data_ptr(), count,
                                                 ops._opt_ptr(g32), ops._opt_ptr(b32), eps, momentum, ops._opt_ptr(running_mean),
                                                 ops._opt_ptr(running_var), coef.data_ptr(), ws.data_ptr(), ws_bytes, stream),
                  'stp3_dwconv2d_fwd_stats_bn')
        scale_p, shift_p = coef.data_ptr(), coef.data_ptr() + 4 * c
        pooled_sum = torch.empty(n, c, dtype=torch.float32, device=dev)
        w1f, w2f = w1.detach().flatten(1).float().contiguous(), w2.detach().flatten(1).float().contiguous()
        md = _lib.SeMlpDims(n, c, w1f.shape[0], 1.0 / float(ho * wo))
        z1 = torch.empty(n, md.S, dtype=torch.float32, device=dev)
        gate = torch.empty(n, c, dtype=torch.float32, device=dev)
        check(lib.stp3_se_pool_act(ctypes.byref(sd), e2.data_ptr(), scale_p, shift_p, ops.ACT_SWISH, ws.data_ptr(), ws_bytes,
                                   pooled_sum.data_ptr(), stream), 'stp3_se_pool_act')
        check(lib.stp3_se_mlp_fwd(ctypes.byref(md), pooled_sum.data_ptr(), w1f.data_ptr(), ops._f32(b1).data_ptr(),
                                  w2f.data_ptr(), ops._f32(b2).data_ptr(), z1.data_ptr(), gate.data_ptr(), stream),
              'stp3_se_mlp_fwd')
        a = torch.empty_like(e2)
        check(lib.stp3_mbconv_scale_act(ctypes.byref(sd), c, e2.data_ptr(), scale_p, shift_p, ops.ACT_SWISH, gate.data_ptr(),
                                        a.data_ptr(), stream), 'stp3_mbconv_scale_act')
        ctx.save_for_backward(x, wt, e2, coef, gate, pooled_sum, z1, w1f, w2f)
        ctx.cfg = (dwd, sd, md, count, exchange, group, ws_bytes)
        ctx.meta = (dw_weight.shape, dw_weight.dtype, None if gamma is None else gamma.dtype,
                    None if beta is None else beta.dtype, w1.shape, w2.shape, w1.dtype, b1.dtype, w2.dtype, b2.dtype)
        return a

    @staticmethod
    def backward(ctx, da):
        x, wt, e2, coef, gate, pooled_sum, z1, w1f, w2f = ctx.saved_tensors
        dwd, sd, md, count, exchange, group, ws_bytes = ctx.cfg
        wshape, wdt, gdt, bdt, w1s, w2s, w1d, b1d, w2d, b2d = ctx.meta
        lib = _lib.lib()
        dev = x.device
        n, c = sd.N, sd.C
        if da.dtype != e2.dtype:
            da = da.to(e2.dtype)
        da = da.contiguous(memory_format=torch.channels_last)
        ws = _workspace(ws_bytes, dev)
        stream = ops._stream_handle()
        f32 = dict(dtype=torch.float32, device=dev)
        sums5 = torch.empty(5, n, c, **f32)
        dz2, dz1 = torch.empty(n, c, **f32), torch.empty(n, md.S, **f32)
        dpooled = torch.empty(n, c, **f32)
        dw1, db1 = torch.empty(md.S, c, **f32), torch.empty(md.S, **f32)
        dw2, db2 = torch.empty(c, md.S, **f32), torch.empty(c, **f32)
        check(lib.stp3_mbconv_bwd_reduce(ctypes.byref(sd), c, da.data_ptr(), e2.data_ptr(), coef.data_ptr(), ops.ACT_SWISH,
                                         ws.data_ptr(), ws_bytes, sums5.data_ptr(), stream), 'stp3_mbconv_bwd_reduce')
        check(lib.stp3_se_mlp_bwd(ctypes.byref(md), sums5.data_ptr(), gate.data_ptr(), pooled_sum.data_ptr(), z1.data_ptr(),
                                  w1f.data_ptr(), w2f.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), dpooled.data_ptr(),
                                  dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), stream), 'stp3_se_mlp_bwd')
        lsums = torch.empty(2, c, **f32)                                  # sum g (dbeta) | sum g * xhat (dgamma), this rank
        check(lib.stp3_mbconv_bwd_coef(n, c, sums5.data_ptr(), gate.data_ptr(), dpooled.data_ptr(), lsums.data_ptr(), stream),
              'stp3_mbconv_bwd_coef')
        gsums = lsums
        if exchange:
            gsums = lsums.clone()
            torch.distributed.all_reduce(gsums, group=group)
            ops._EXCHANGES['batchnorm'] += 1
        de2 = torch.empty_like(e2)
        check(lib.stp3_mbconv_bwd_apply(ctypes.byref(sd), c, da.data_ptr(), e2.data_ptr(), coef.data_ptr(), ops.ACT_SWISH,
                                        gate.data_ptr(), dpooled.data_ptr(), gsums.data_ptr(), count, de2.data_ptr(), stream),
              'stp3_mbconv_bwd_apply')
        dx = dwg = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            check(lib.stp3_dwconv2d_bwd_data(ctypes.byref(dwd), de2.data_ptr(), wt.data_ptr(), dx.data_ptr(), stream),
                  'stp3_dwconv2d_bwd_data')
        if ctx.needs_input_grad[1]:
            nbytes = ctypes.c_size_t()
            check(lib.stp3_dwconv2d_bwd_weight_workspace(ctypes.byref(dwd), ctypes.byref(nbytes)),
                  'stp3_dwconv2d_bwd_weight_workspace')
            ws2 = _workspace(max(nbytes.value, ws_bytes), dev)
            # (in the parameter's own layout: autograd takes the tensor as the gradient as it is -- the tap-major form needed a
            # transposing copy per layer and step)
            dwg = torch.empty(wshape, dtype=torch.float32, device=dev)
            check(lib.stp3_dwconv2d_bwd_weight_oihw(ctypes.byref(dwd), x.data_ptr(), de2.data_ptr(), dwg.data_ptr(), ws2.data_ptr(),
                                                    nbytes.value, stream), 'stp3_dwconv2d_bwd_weight_oihw')
            dwg = dwg.to(wdt)
        dgamma = lsums[1].to(gdt) if gdt is not None and ctx.needs_input_grad[4] else None
        dbeta = lsums[0].to(bdt) if bdt is not None and ctx.needs_input_grad[5] else None
        return (dx, dwg, None, None, dgamma, dbeta, None, None, None, None, dw1.view(w1s).to(w1d), db1.to(b1d),
                dw2.view(w2s).to(w2d), db2.to(b2d), None)


def dw_bn_se_supported(x, dw_conv, bn):
    """The fused MBConv middle takes GPU bf16 / float32 activations in whole 16-byte channel vectors, training-mode
    BatchNorm with running statistics, 3x3 / 5x5 depthwise kernels with stride 1 / 2."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32)):
        return False
    per = 8 if x.dtype == torch.bfloat16 else 4
    return (bn.training and bn.track_running_stats and x.shape[1] % per == 0 and dw_conv.kernel_size[0] in (3, 5)
            and dw_conv.stride[0] in (1, 2))


def dw_bn_se(x, dw_conv, bn, se_reduce, se_expand, group=None):
    """``se(swish(bn(dw_conv(x))))`` of an MBConv block through ``_DwBnSe``; ``dw_conv``: the block's
    ``StaticSamePadConv2d`` depthwise module (its frozen padding is applied in-kernel)."""
    if bn.num_batches_tracked is not None:
        ops.bump_batch_counter(bn)
    return _DwBnSe.apply(x, dw_conv.weight, int(dw_conv.stride[0]), tuple(int(p) for p in dw_conv._pad), bn.weight, bn.bias,
                         bn.running_mean, bn.running_var, ops.bn_momentum(bn),
                         float(bn.eps), se_reduce.weight, se_reduce.bias, se_expand.weight, se_expand.bias, group)
