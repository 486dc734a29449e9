"""EfficientNet trunk (B0 / B4) as the reference's ``Encoder`` uses it.

The reference imports ``efficientnet_pytorch==0.7.0`` (environment.yml:18), which is not vendored
under /root/reference and not installed here; this file restates the published architecture
(Tan & Le 2019) behind the attribute names the reference touches
(stp3/models/encoder.py:18, 41-55, 62-70): ``_conv_stem, _bn0, _swish, _blocks[i](x,
drop_connect_rate=...), _global_params.drop_connect_rate`` and the parameter names of that
package, so its checkpoints load (``_blocks.N._expand_conv / _bn0 / _depthwise_conv / _bn1 /
_se_reduce / _se_expand / _project_conv / _bn2``).  PARITY UNPINNED against the real package
(its source is absent); parity is pinned only in the sense that the oracle plugs this very trunk
into the reference's own ``Encoder`` class.

"Static same" padding: the package fixes each conv's padding for the canonical resolution of
the variant (380 px for B4, 224 px for B0), asymmetric (extra pixel right/bottom) on stride-2
convs, and applies it unchanged to the 224x480 inputs used here.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..layers import fused as _fused
from ..layers.fused import ACT_NONE, ACT_SWISH, RES_AFTER_ACT, bn_act, conv2d


# (repeats, kernel, stride, expand, in, out) of the 7 base stages, SE ratio 0.25 everywhere
_BASE_STAGES = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
                (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))
# name -> (width multiplier, depth multiplier, canonical resolution, dropout)
_VARIANTS = {'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b4': (1.4, 1.8, 380, 0.4)}


def _round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def _round_repeats(repeats, depth):
    return int(math.ceil(depth * repeats))


class StaticSamePadConv2d(nn.Conv2d):
    """Conv2d whose TF-"same" padding is frozen for a given input size (square ``image_size``)."""

    def __init__(self, in_ch, out_ch, kernel_size, image_size, stride=1, groups=1, bias=False):
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, groups=groups, bias=bias)
        k, s = self.kernel_size[0], self.stride[0]
        out = math.ceil(image_size / s)
        pad = max((out - 1) * s + (k - 1) + 1 - image_size, 0)
        self._pad = (pad // 2, pad - pad // 2, pad // 2, pad - pad // 2)   # left, right, top, bottom

    def forward(self, x):
        if self.groups == self.in_channels == self.out_channels and self.groups > 1 and x.is_cuda and x.dtype != torch.float64:
            # depthwise: hand-written HIP kernels (padding handled in-kernel, no F.pad copy)
            return ops.depthwise_conv2d(x, self.weight, self.stride[0], self._pad)
        if self.kernel_size == (1, 1) and x.shape[-2:] == (1, 1) and self.stride == (1, 1):
            # squeeze-excite 1x1 convs on a pooled 1x1 map are plain GEMMs
            y = F.linear(x.flatten(1), self.weight.flatten(1), self.bias)
            return y.view(*y.shape, 1, 1)
        if self.groups == 1 and x.is_cuda and self.in_channels % 8 and (torch.is_autocast_enabled() or x.dtype == torch.float32):
            # the 3-channel stem: zero-pad the channels to 8 (together with the "same" padding, one copy) so that
            # it runs on the MFMA kernel too; the padded weight columns are zero and their gradient is dropped
            cp = (-self.in_channels) % 8
            if x.requires_grad:
                x = F.pad(x, (*self._pad, 0, cp))
            else:
                # the camera images: ONE strided copy (with the cast) into the zeroed channels-last bf16 operand,
                # instead of a float32 pad, a cast and a layout change (0.6 ms per step)
                n, c, h, w = x.shape
                left, right, top, bottom = self._pad
                xp = torch.empty((n, c + cp, h + top + bottom, w + left + right),
                                 dtype=torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype,
                                 device=x.device, memory_format=torch.channels_last).zero_()
                xp[:, :c, top:top + h, left:left + w] = x
                x = xp
            if self.out_channels % 8 == 0 and ops.assembled_weight_supported(x, (self.weight,)):
                # (the weight with its zero columns: an assembled weight -- no pad per step, no slice of its gradient)
                wp = ops.assembled_weight((id(self), 'padded'), (self.out_channels, self.in_channels + cp, *self.kernel_size),
                                          [ops.weight_piece(self.weight, self.weight.detach())])
            else:
                wp = F.pad(self.weight, (0, 0, 0, 0, 0, cp))
            return conv2d(x, wp, self.bias, self.stride, 0, self.dilation)
        if any(self._pad):
            x = F.pad(x, self._pad)
        if self.groups == 1:
            return conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation)
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


# expand convolution -> BN0 -> swish through ops_fused._PointwiseBnAct (the convolution output is recomputed, never stored)
EXPAND_WITHOUT_E0 = True
# the gradient of a block's identity skip added by the expand convolution's data-gradient kernel (ops.SkipCarrier)
SKIP_GRADIENT_IN_DGRAD = True


class Swish(nn.Module):
    def forward(self, x):
        return F.silu(x)


class MBConvBlock(nn.Module):
    def __init__(self, in_ch, out_ch, kernel, stride, expand, image_size, se_ratio=0.25, bn_mom=0.01, bn_eps=1e-3):
        super().__init__()
        self.in_ch, self.out_ch, self.stride, self.expand = in_ch, out_ch, stride, expand
        mid = in_ch * expand
        if expand != 1:
            self._expand_conv = StaticSamePadConv2d(in_ch, mid, 1, image_size)
            self._bn0 = nn.BatchNorm2d(mid, momentum=bn_mom, eps=bn_eps)
        self._depthwise_conv = StaticSamePadConv2d(mid, mid, kernel, image_size, stride=stride, groups=mid)
        self._bn1 = nn.BatchNorm2d(mid, momentum=bn_mom, eps=bn_eps)
        out_size = math.ceil(image_size / stride)
        squeezed = max(1, int(in_ch * se_ratio))
        self._se_reduce = StaticSamePadConv2d(mid, squeezed, 1, 1, bias=True)
        self._se_expand = StaticSamePadConv2d(squeezed, mid, 1, 1, bias=True)
        self._project_conv = StaticSamePadConv2d(mid, out_ch, 1, out_size)
        self._bn2 = nn.BatchNorm2d(out_ch, momentum=bn_mom, eps=bn_eps)
        self._swish = Swish()
        self.out_size = out_size

    def forward(self, inputs, drop_connect_rate=None, drop_scale=None):
        """``drop_scale`` (N,): this block's drop-connect factors mask_n / keep drawn by the caller (``Encoder.trunk``
        draws those of all blocks together); otherwise they are drawn here from ``drop_connect_rate``."""
        x = inputs
        # 1x1 conv -> BN -> activation as one operator (BatchNorm statistics from the convolution epilogue)
        fuse = (self.training and x.is_cuda and torch.is_autocast_enabled()
                and self.in_ch % 8 == 0 and (self.in_ch * self.expand) % 8 == 0
                # the statistics epilogue of the convolution kernel reduces at most 65 536 row tiles of 128 pixels
                and (x.shape[0] * x.shape[2] * x.shape[3] + 127) // 128 <= ops._CONV_MAX_STAT_TILES)
        carrier = None
        if fuse:
            from .. import ops_fused
            group = None if _fused._sync_world(self._bn1) > 1 else False
            # the identity skip's gradient travels from the project BatchNorm (which adds the skip) to the expand convolution
            # (which consumes the block input) and joins its data gradient in the kernel that writes it, instead of being
            # added by the autograd engine in a pass of its own (bf16 blocks with an expand layer and an input gradient)
            if (SKIP_GRADIENT_IN_DGRAD and self.expand != 1 and self.stride == 1 and self.in_ch == self.out_ch
                    and x.requires_grad and torch.is_grad_enabled() and x.dtype == torch.bfloat16
                    and x.is_contiguous(memory_format=torch.channels_last)):
                carrier = ops.SkipCarrier()
        if self.expand != 1:
            if fuse and EXPAND_WITHOUT_E0 and ops_fused.pointwise_bn_act_supported(x, self._expand_conv, self._bn0) \
                    and ops_fused.pointwise_bn_act_pays(x, self._expand_conv):
                # the expanded pre-activation tensor (6x the block input) is never stored: statistics pass + recomputation
                x = ops_fused.pointwise_bn_act(x, self._expand_conv, self._bn0, ACT_SWISH, group=group, skip_carrier=carrier)
            elif fuse:
                x = ops_fused.conv_bn_act(x, self._expand_conv.weight, None, self._bn0, ACT_SWISH, group=group,
                                          skip_carrier=carrier)
            else:
                x = bn_act(self._bn0, self._expand_conv(x), ACT_SWISH)
        from .. import ops_fused
        if x.is_cuda and torch.is_autocast_enabled() and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype('cuda'))
        if self.training and ops_fused.dw_bn_se_supported(x, self._depthwise_conv, self._bn1):
            # depthwise -> BN1 -> swish -> squeeze-excite as one operator: the BatchNorm statistics come from the
            # depthwise kernel's epilogue and swish(BN1(.)) is applied where it is consumed, never written
            x = ops_fused.dw_bn_se(x, self._depthwise_conv, self._bn1, self._se_reduce, self._se_expand,
                                   group=None if _fused._sync_world(self._bn1) > 1 else False)
        elif x.is_cuda and x.dtype != torch.float64:
            x = bn_act(self._bn1, self._depthwise_conv(x), ACT_SWISH)
            x = ops_fused.se_block(x, self._se_reduce, self._se_expand)
        else:
            x = bn_act(self._bn1, self._depthwise_conv(x), ACT_SWISH)
            s = x.mean((2, 3), keepdim=True)
            s = self._se_expand(self._swish(self._se_reduce(s)))
            x = torch.sigmoid(s) * x
        skip = self.stride == 1 and self.in_ch == self.out_ch
        oscale = drop_scale if skip else None
        if oscale is None and skip and drop_connect_rate and self.training:
            # drop-connect: the branch of sample n is scaled by mask_n / keep before the skip is added
            keep = 1.0 - drop_connect_rate
            oscale = torch.floor(keep + torch.rand(x.shape[0], dtype=torch.float32, device=x.device)) / keep
        if fuse:
            return ops_fused.conv_bn_act(x, self._project_conv.weight, None, self._bn2, ACT_NONE, inputs if skip else None,
                                         RES_AFTER_ACT if skip else _fused.RES_NONE, group=group, oscale=oscale,
                                         res_carrier=carrier if skip else None)
        x = self._project_conv(x)
        if skip:
            return bn_act(self._bn2, x, ACT_NONE, res=inputs, res_mode=RES_AFTER_ACT, oscale=oscale)
        return bn_act(self._bn2, x, ACT_NONE)


class EfficientNet(nn.Module):
    """Trunk + (unused here) head placeholders so that the reference's ``delete_unused_layers``
    (encoder.py:39-55) finds the attributes it deletes."""

    def __init__(self, name='efficientnet-b4', drop_connect_rate=0.2):
        super().__init__()
        width, depth, res, dropout = _VARIANTS[name]
        self._global_params = SimpleNamespace(drop_connect_rate=drop_connect_rate, image_size=res,
                                              width_coefficient=width, depth_coefficient=depth)
        stem = _round_filters(32, width)
        self._conv_stem = StaticSamePadConv2d(3, stem, 3, res, stride=2)
        self._bn0 = nn.BatchNorm2d(stem, momentum=0.01, eps=1e-3)
        size = math.ceil(res / 2)
        blocks = []
        for repeats, k, s, e, i, o in _BASE_STAGES:
            i, o = _round_filters(i, width), _round_filters(o, width)
            for r in range(_round_repeats(repeats, depth)):
                blk = MBConvBlock(i if r == 0 else o, o, k, s if r == 0 else 1, e, size)
                size = blk.out_size
                blocks.append(blk)
        self._blocks = nn.ModuleList(blocks)
        head = _round_filters(1280, width)
        self._conv_head = StaticSamePadConv2d(blocks[-1].out_ch, head, 1, size)
        self._bn1 = nn.BatchNorm2d(head, momentum=0.01, eps=1e-3)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(dropout)
        self._fc = nn.Linear(head, 1000)
        self._swish = Swish()

    @classmethod
    def from_name(cls, name):
        return cls(name)

    @classmethod
    def from_pretrained(cls, name):
        """The package would download ImageNet weights here; there is no network, so this is a
        seeded random initialisation of the same architecture (load real weights with
        ``load_state_dict`` -- the parameter names match).  Said out loud, once: a silently random backbone is the
        kind of surprise that costs a training run."""
        import warnings
        warnings.warn(f'EfficientNet.from_pretrained({name!r}): no ImageNet weights are available offline -- the '
                      f'backbone is RANDOMLY initialised; load real weights with load_state_dict()', stacklevel=2)
        return cls(name)
