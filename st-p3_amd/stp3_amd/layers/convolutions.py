"""2-D building blocks on the LSS path, parameter-name compatible with the reference's
``stp3/layers/convolutions.py`` (UpsamplingConcat :183-201, UpsamplingAdd :204-215, ASPP
:217-270, DeepLabHead :272-280).  The off-path blocks of that file (ConvBlock, Bottleneck,
ConvNeXt Block, ...) are out of scope.

Exact work reductions relative to a literal translation (no approximation):
  * a dilated 3x3 whose dilation is >= the feature map's height/width only ever reads its
    centre row/column/tap (all other taps fall in the zero padding); ``ASPPConv`` then slices the
    weight and runs the smaller convolution;
  * the ASPP image-pooling branch is spatially constant; it is applied as a per-sample bias of
    the 1x1 projection instead of being materialised, interpolated and concatenated.
"""
import torch
import torch.nn as nn

from .. import ops, ops_pred
from ..utils import hp
from .fused import (ACT_NONE, ACT_RELU, RES_AFTER_ACT, _sync_world, bn_act, bn_act_group, conv1x1_on_vector, conv2d, conv_bn_act_layer,
                    conv_bn_act_member, conv_module, plane_mean, pooled_bias, run_fused)
from .fused import _emu as fused_emu


def _conv_bn_relu(in_ch, out_ch, k, padding=0, dilation=1):
    return [nn.Conv2d(in_ch, out_ch, k, padding=padding, dilation=dilation, bias=False), nn.BatchNorm2d(out_ch),
            nn.ReLU(inplace=True)]


def _upsample(module, x, dtype):
    """``module(x)`` for an ``nn.Upsample``, in ``dtype``.  Bilinear integer-factor up-sampling of GPU tensors runs on
    the kernels of stp3_upsample.hip in the tensor's own type (float32 arithmetic in torch's order, one rounding).
    Anything else takes torch's operator, which under autocast interpolates in float32 -- that result is rounded once
    here instead of being concatenated / consumed as float32."""
    scale = module.scale_factor
    if isinstance(scale, (tuple, list)) and len(set(scale)) == 1:
        scale = scale[0]
    if (x.is_cuda and module.mode == 'bilinear' and not module.align_corners and module.size is None
            and isinstance(scale, (int, float))):
        xin = x.to(dtype) if x.dtype != dtype and torch.is_autocast_enabled() else x
        if ops.upsample_bilinear_supported(xin, scale):
            return fused_emu(ops.upsample_bilinear(xin, scale))
    y = module(x)
    return y.to(dtype) if y.dtype != dtype and torch.is_autocast_enabled() else y


class UpsamplingConcat(nn.Module):
    """x2 bilinear upsample, concat [skip, upsampled], 2 x (3x3 conv + BN + ReLU)."""

    def __init__(self, in_channels, out_channels, scale_factor=2):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=False)
        self.conv = nn.Sequential(*_conv_bn_relu(in_channels, out_channels, 3, padding=1),
                                  *_conv_bn_relu(out_channels, out_channels, 3, padding=1))

    def forward(self, x_to_upsample, x):
        joined = self._joined(x_to_upsample, x)
        if joined is None:
            joined = torch.cat([x, _upsample(self.upsample, x_to_upsample, x.dtype)], dim=1)
        return run_fused(self.conv, joined)

    def _joined(self, x_to_upsample, x):
        """[skip, upsampled] WITHOUT the concatenation: the up-sampling kernel writes its rows into its channel slice of the
        3x3 convolution's operand, the skip is copied into the other (one pass over the smaller tensor instead of
        torch.cat's pass over both); bf16 GPU tensors with 16-byte channel offsets only."""
        from .fused import slot_ok
        up = self.upsample
        scale = up.scale_factor[0] if isinstance(up.scale_factor, (tuple, list)) and len(set(up.scale_factor)) == 1 else up.scale_factor
        if not (slot_ok(x) and x_to_upsample.is_cuda and up.mode == 'bilinear' and not up.align_corners and up.size is None
                and isinstance(scale, (int, float)) and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last)):
            return None
        xin = x_to_upsample.to(x.dtype) if x_to_upsample.dtype != x.dtype and torch.is_autocast_enabled() else x_to_upsample
        if not (xin.dtype == x.dtype and ops.upsample_bilinear_supported(xin, scale)
                and (xin.shape[2] * int(scale), xin.shape[3] * int(scale)) == tuple(x.shape[2:])):
            return None
        from .. import ops_fused
        n, cs, h, w = x.shape
        buf = torch.empty((n, cs + xin.shape[1], h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        parts = [ops_fused.copy_into_slot(x, (buf, 0)), ops.upsample_bilinear(xin, scale, out_slot=(buf, cs))]
        return ops_fused.join_slices(buf, parts)


class UpsamplingAdd(nn.Module):
    """x2 bilinear upsample, 1x1 conv + BN, add the skip."""

    def __init__(self, in_channels, out_channels, scale_factor=2):
        super().__init__()
        self.upsample_layer = nn.Sequential(
            nn.Upsample(scale_factor=scale_factor, mode='bilinear', align_corners=False),
            nn.Conv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels))

    def forward(self, x, x_skip):
        up, conv, bn = self.upsample_layer
        return conv_bn_act_layer(_upsample(up, x, x.dtype), conv, bn, ACT_NONE, res=x_skip, res_mode=RES_AFTER_ACT)


class ASPPConv(nn.Sequential):
    """3x3 dilated conv + BN + ReLU (modules 0,1,2 like the reference's Sequential)."""

    def __init__(self, in_channels, out_channels, dilation):
        super().__init__(nn.Conv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU())
        self.dilation = dilation

    def conv_only(self, x):
        """The convolution of the branch, with the taps that can never be in range dropped (exact)."""
        conv = self[0]
        h, w = x.shape[-2:]
        d = self.dilation
        wgt = conv.weight
        rows, cols = (slice(1, 2) if d >= h else slice(None)), (slice(1, 2) if d >= w else slice(None))
        if d < h and d < w:
            return conv_module(conv, x)
        if wgt.shape[0] % 8 == 0 and x.shape[1] % 8 == 0 and ops.assembled_weight_supported(x, (wgt,)):
            # the taps that remain as a weight of their own, cut from the parameter by the launch that refreshes every bf16
            # shadow (ops.assembled_weight: no slice / cast / re-layout per step, and the gradient goes back the same way)
            kept = wgt.detach()[:, :, rows, cols]
            sub = ops.assembled_weight((id(conv), 'kept taps'), tuple(kept.shape), [ops.weight_piece(wgt, kept)])
        else:
            sub = wgt[:, :, rows, cols]
        if d >= h and d >= w:                       # only the centre tap can ever be in range
            return conv2d(x, sub)
        if d >= h:                                  # centre row only: 1x3
            return conv2d(x, sub, padding=(0, d), dilation=(1, d))
        return conv2d(x, sub, padding=(d, 0), dilation=(d, 1))          # centre column only: 3x1

    def forward(self, x):
        h, w = x.shape[-2:]
        if self.dilation < h and self.dilation < w:
            # no tap dropped: the plain conv -> BN -> ReLU chain, i.e. the operator whose convolution epilogue already
            # produces the BatchNorm statistics (one pass over the branch output less)
            return run_fused(self, x)
        return bn_act(self[1], self.conv_only(x), ACT_RELU)


class ASPPPooling(nn.Sequential):
    """Global average pool -> 1x1 conv + BN + ReLU (modules 0..3 like the reference)."""

    def __init__(self, in_channels, out_channels):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU())

    def conv_only(self, x):
        pool, conv, _, _ = self
        # the global average through fused.plane_mean: same value as AdaptiveAvgPool2d(1), gradient in the layout of x
        pooled = plane_mean(x).to(x.dtype)[:, :, None, None] if x.is_cuda else pool(x)
        return conv_module(conv, pooled)            # 1x1 map: a GEMM

    def forward(self, x):
        """Returns the (N, C, 1, 1) pooled descriptor; bilinear upsampling of a 1x1 map is a
        broadcast, which ``ASPP`` folds into its projection."""
        return bn_act(self[2], self.conv_only(x), ACT_RELU)   # the fused BatchNorm on N vectors


class ASPP(nn.Module):
    def __init__(self, in_channels, atrous_rates, out_channels=256):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels),
                              nn.ReLU())]
        mods += [ASPPConv(in_channels, out_channels, r) for r in tuple(atrous_rates)]
        mods.append(ASPPPooling(in_channels, out_channels))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(len(self.convs) * out_channels, out_channels, 1, bias=False),
                                     nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(0.5))

    def _branches_into_one_buffer(self, xs):
        """The spatial branches (1x1 and the dilated 3x3s, each conv -> BN -> ReLU) with their results written straight into
        the channel slices of ONE buffer -- the concatenation the projection reads -- instead of four tensors and a
        ``torch.cat`` (491 MB written and read again for the temporal head of configs[2]).  None when a branch does not
        qualify for the fused operator (dropped taps, evaluation mode, float32): then the plain route is taken."""
        from .fused import _fusable_conv_bn, _sync_world as sync_world
        from .. import ops_fused
        convs = [(m[0], m[1]) for m in self.convs[:-1]]
        h, w = xs[0].shape[-2:]
        if not all(_fusable_conv_bn(c, bn, xs[0]) for c, bn in convs):
            return None
        if any(isinstance(m, ASPPConv) and not (m.dilation < h and m.dilation < w) for m in self.convs[:-1]):
            return None
        co = convs[0][0].out_channels
        if co % 8 or any(c.out_channels != co for c, _ in convs):
            return None
        if xs[0].dtype != torch.bfloat16:
            return None
        n = xs[0].shape[0]
        buf = torch.empty((n, co * len(convs), h, w), dtype=torch.bfloat16, device=xs[0].device, memory_format=torch.channels_last)
        parts = []
        for k, ((c, bn), xi) in enumerate(zip(convs, xs)):
            parts.append(ops_fused.conv_bn_act(xi, c.weight, c.bias, bn, ACT_RELU, None, ops.RES_NONE, c.stride, c.padding,
                                               c.dilation, group=False, out_slot=(buf, k * co)))
        return ops_fused.join_slices(buf, parts)

    def forward(self, x):
        # the branches read one tensor: their input gradients are added in one pass (ops.fan_out), not pairwise
        xs = ops.fan_out(x, len(self.convs)) if x.is_cuda else [x] * len(self.convs)
        spatial = None
        if _sync_world(self.convs[0][1]) > 1:
            # N > 1 ranks: the five branches are siblings -- their BatchNorm statistics travel in ONE exchange per pass
            members = [conv_bn_act_member(xs[0], self.convs[0][0], self.convs[0][1], ACT_RELU)]
            members += [dict(bn=conv[1], x=conv.conv_only(xi), act=ACT_RELU) for conv, xi in zip(self.convs[1:-1], xs[1:-1])]
            members.append(dict(bn=self.convs[-1][2], x=self.convs[-1].conv_only(xs[-1]), act=ACT_RELU))
            *branches, pooled = bn_act_group(members)
        else:
            spatial = self._branches_into_one_buffer(xs)
            if spatial is None:
                branches = [run_fused(self.convs[0], xs[0])] + [conv(xi) for conv, xi in zip(self.convs[1:-1], xs[1:-1])]
            pooled = self.convs[-1](xs[-1])                          # (N, C, 1, 1)
        if spatial is None:
            spatial = torch.cat(branches, dim=1)
        proj, bn, act, drop = self.project
        n_sp = spatial.shape[1]
        # (split, not two slices: one concatenation in backward instead of two zero-fills, two copies and an addition)
        if n_sp % 8 == 0 and proj.weight.shape[0] % 8 == 0 and ops.assembled_weight_supported(spatial, (proj.weight,)):
            # the columns of the spatial branches as a weight of their own (ops.assembled_weight); the pooled branch's columns
            # go another way (below) and write their part of the parameter's gradient themselves (ops.weight_columns)
            w_sp = ops.assembled_weight((id(proj), 'spatial'), (proj.weight.shape[0], n_sp, 1, 1),
                                        [ops.weight_piece(proj.weight, proj.weight.detach()[:, :n_sp])], direct='shared')
            w_pool = ops.weight_columns(w_sp, proj.weight, n_sp, proj.weight.shape[1])
        else:
            w_sp, w_pool = proj.weight.split([n_sp, proj.weight.shape[1] - n_sp], dim=1)
        y = conv2d(spatial, w_sp)
        # the pooled branch is a constant plane per sample: its projection is a per-sample bias, folded
        # into the fused BatchNorm instead of a broadcast add over the whole map
        sbias = pooled_bias(hp(conv1x1_on_vector(pooled, w_pool).flatten(1)))
        return drop(bn_act(bn, y, ACT_RELU, sbias=sbias))


class DeepLabHead(nn.Sequential):
    def __init__(self, in_channels, num_classes, hidden_channel=256):
        super().__init__(ASPP(in_channels, [12, 24, 36], hidden_channel),
                         nn.Conv2d(hidden_channel, hidden_channel, 3, padding=1, bias=False),
                         nn.BatchNorm2d(hidden_channel), nn.ReLU(), nn.Conv2d(hidden_channel, num_classes, 1))

    def forward(self, x):
        return run_fused(self, x)


# ----------------------------------------------------------------------------------------------
# Blocks of the prediction stage: stp3/layers/convolutions.py:10-171 (ConvBlock, Bottleneck),
# :283-380 (LayerNorm, ConvNeXt Block, Bottleblock).  Same constructors, attributes and parameter names as the
# reference's; dense convolutions and BatchNorm + ReLU pairs go through the same operators as the perception path
# (``conv_module`` -> MFMA implicit GEMM under bf16 autocast, ``run_fused`` -> stp3_bn_*), the 7x7 depthwise
# convolution, LayerNorm and GELU are torch operators for now.
# ----------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402
from collections import OrderedDict  # noqa: E402


class Bottleneck(nn.Module):
    """1x1 down-projection -> kxk (optionally stride-2) convolution -> 1x1 up-projection, BatchNorm + ReLU after each,
    plus the (projected / max-pooled) skip: stp3/layers/convolutions.py:62-171.  The transposed-convolution up-sampling
    variant is not used on ST-P3's prediction path and is rejected."""

    def __init__(self, in_channels, out_channels=None, kernel_size=3, dilation=1, groups=1, upsample=False,
                 downsample=False, dropout=0.0):
        super().__init__()
        if upsample:
            raise NotImplementedError('Bottleneck(upsample=True) is not on the prediction path')
        assert dilation == 1
        self._downsample = downsample
        mid = int(in_channels / 2)
        out_channels = out_channels or in_channels
        pad = ((kernel_size - 1) * dilation + 1) // 2
        conv = nn.Conv2d(mid, mid, kernel_size=kernel_size, bias=False, dilation=dilation, stride=2 if downsample else 1,
                         padding=pad, groups=groups)
        self.layers = nn.Sequential(OrderedDict([
            ('conv_down_project', nn.Conv2d(in_channels, mid, kernel_size=1, bias=False)),
            ('abn_down_project', nn.Sequential(nn.BatchNorm2d(mid), nn.ReLU(inplace=True))),
            ('conv', conv),
            ('abn', nn.Sequential(nn.BatchNorm2d(mid), nn.ReLU(inplace=True))),
            ('conv_up_project', nn.Conv2d(mid, out_channels, kernel_size=1, bias=False)),
            ('abn_up_project', nn.Sequential(nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))),
            ('dropout', nn.Dropout2d(p=dropout))]))
        if out_channels == in_channels and not downsample:
            self.projection = None
        else:
            proj = OrderedDict()
            if downsample:
                proj['upsample_skip_proj'] = nn.MaxPool2d(kernel_size=2, stride=2)
            proj['conv_skip_proj'] = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
            proj['bn_skip_proj'] = nn.BatchNorm2d(out_channels)
            self.projection = nn.Sequential(proj)

    def forward(self, x):
        y = run_fused(self.layers, x)
        if self.projection is None:
            return y + x
        if self._downsample:
            x = F.pad(x, (0, x.shape[-1] % 2, 0, x.shape[-2] % 2), value=0)
        return y + run_fused(self.projection, x)


class LayerNorm(nn.Module):
    """LayerNorm over the channels in channels_last (N,H,W,C) or channels_first (N,C,H,W) data (convolutions.py:283-307)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format='channels_last'):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if data_format not in ('channels_last', 'channels_first'):
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == 'channels_last':
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return self.channels_first(x)

    def channels_first(self, x, act=ops_pred.ACT_NONE):
        """The normalisation over dim 1 of (N,C,H,W) [+ GELU]: on the GPU one pass of stp3_layernorm_fwd over the
        channels-last rows; otherwise F.layer_norm on the (N,H,W,C) VIEW, which for the channels-last tensors of this code
        base is contiguous memory (one fused layer_norm instead of the reference's five passes)."""
        if ops_pred.layer_norm_supported(x, self.normalized_shape[0]):
            return ops_pred.layer_norm_channels(x, self.weight, self.bias, self.eps, act)
        y = F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)
        return F.gelu(y) if act == ops_pred.ACT_GELU else y


class Block(nn.Module):
    """ConvNeXt block: 7x7 depthwise -> LayerNorm -> Linear 4x -> GELU -> Linear -> layer scale, + skip
    (convolutions.py:309-345)."""

    def __init__(self, dim, drop_path=0.0, layer_scale_init_value=1e-6):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = (nn.Parameter(layer_scale_init_value * torch.ones(dim), requires_grad=True)
                      if layer_scale_init_value > 0 else None)
        if drop_path > 0.0:
            raise NotImplementedError('drop_path > 0 (timm DropPath) is not used by ST-P3')
        self.drop_path = nn.Identity()

    def forward(self, x):
        dw = self.dwconv
        # (the kernel route is the module's own layer only: 7x7, stride 1, `same` padding, one group per channel)
        on_kernels = (x.is_cuda and tuple(dw.kernel_size) == (7, 7) and tuple(dw.stride) == (1, 1) and tuple(dw.padding) == (3, 3)
                      and tuple(dw.dilation) == (1, 1) and dw.groups == x.shape[1] == dw.out_channels
                      and ops.depthwise_supported(x, dw.weight, 1))
        if on_kernels:
            # the 7x7 depthwise layer on stp3_dwconv2d_* (torch hands it to a naive MIOpen kernel: 1.8 ms forward and
            # 4.3 ms backward per call at 28 x 64 x 200 x 200); channels-last memory, so the (N,H,W,C) view below is free
            y = ops.depthwise_conv2d(x, dw.weight, 1, (3, 3, 3, 3), bias=dw.bias)
        else:
            y = dw(x)
        if on_kernels and (y.dtype == torch.bfloat16 or torch.is_autocast_enabled()):
            # the two Linear layers act on the channels of every pixel: 1x1 convolutions of the channels-last tensor, on
            # the streaming MFMA kernels (hipBLASLt picks a 26 TFLOP/s kernel for the 1 120 000 x 64 x 256 product)
            z = self.norm.channels_first(y)
            h = self.act(conv2d(z, self.pwconv1.weight[:, :, None, None], self.pwconv1.bias))
            o = conv2d(h, self.pwconv2.weight[:, :, None, None], self.pwconv2.bias)
            if self.gamma is None:
                return x + o
            return torch.addcmul(x, o, self.gamma.to(o.dtype).view(1, -1, 1, 1))
        y = y.permute(0, 2, 3, 1)
        y = self.pwconv2(self.act(self.pwconv1(self.norm(y))))
        if self.gamma is not None:
            y = self.gamma * y
        return x + y.permute(0, 3, 1, 2)


class Bottleblock(nn.Module):
    """7x7 -> 1x1 -> 3x3 convolutions with channels-first LayerNorm + GELU, + (projected) skip (convolutions.py:347-380)."""

    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        mid = int(in_channels / 2)
        out_channels = out_channels or in_channels
        self.layers = nn.Sequential(
            nn.Conv2d(in_channels, mid, kernel_size=7, bias=False, padding=3),
            LayerNorm(mid, eps=1e-6, data_format='channels_first'), nn.GELU(),
            nn.Conv2d(mid, mid, kernel_size=1, bias=False),
            LayerNorm(mid, eps=1e-6, data_format='channels_first'), nn.GELU(),
            nn.Conv2d(mid, out_channels, kernel_size=3, bias=False, padding=1),
            LayerNorm(out_channels, eps=1e-6, data_format='channels_first'), nn.GELU())
        self.projection = None if out_channels == in_channels else nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False), nn.GELU())

    def forward(self, x):
        y, mods, i = x, list(self.layers), 0
        while i < len(mods):                         # conv -> LayerNorm -> GELU three times: the last two as one pass
            m = mods[i]
            if isinstance(m, LayerNorm) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU) \
                    and m.data_format == 'channels_first':
                y = m.channels_first(y, ops_pred.ACT_GELU)
                i += 2
                continue
            y = conv_module(m, y) if type(m) is nn.Conv2d else m(y)
            i += 1
        return y + (x if self.projection is None else run_fused(self.projection, x))
