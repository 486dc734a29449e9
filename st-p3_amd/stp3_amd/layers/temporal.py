"""3-D (time x BEV) blocks on the LSS path, parameter-name compatible with the reference's
``stp3/layers/temporal.py`` (CausalConv3d :252-273, conv_1x1x1_norm_activated :315-325,
PyramidSpatioTemporalPooling :375-423, TemporalBlock :426-489).  The GRU variants and
``Bottleneck3D`` (never built with INBETWEEN_LAYERS=0) are out of scope.

Exact restructuring: the pyramid-pooling branch pools over the whole BEV plane, so its output is
one vector per (sample, frame); instead of upsampling it to X x Y and concatenating, its
contribution to the aggregation conv is added as a per-(sample, frame) bias (SURVEY.md section 7,
item 10).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class CausalConv3d(nn.Module):
    """Conv3d that pads time on the left only (kernel_time - 1 frames), + BN3d + ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size=(2, 3, 3), dilation=(1, 1, 1), bias=False):
        super().__init__()
        assert len(kernel_size) == 3
        kt, kh, kw = kernel_size
        self._tpad = (kt - 1) * dilation[0]
        hp, wp = ((kh - 1) * dilation[1]) // 2, ((kw - 1) * dilation[2]) // 2
        self.pad = nn.ConstantPad3d((wp, wp, hp, hp, self._tpad, 0), 0)   # kept for attribute parity
        self._hw_pad = (0, hp, wp)
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, dilation=dilation, stride=1, padding=0,
                              bias=bias)
        self.norm = nn.BatchNorm3d(out_channels)
        self.activation = nn.ReLU(inplace=True)

    def forward(self, x):
        if self._tpad:
            x = F.pad(x, (0, 0, 0, 0, self._tpad, 0))
        x = F.conv3d(x, self.conv.weight, self.conv.bias, 1, self._hw_pad, self.conv.dilation)
        return self.activation(self.norm(x))


def conv_1x1x1_norm_activated(in_channels, out_channels):
    return nn.Sequential(OrderedDict([('conv', nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False)),
                                      ('norm', nn.BatchNorm3d(out_channels)),
                                      ('activation', nn.ReLU(inplace=True))]))


class PyramidSpatioTemporalPooling(nn.Module):
    """Causal (2-frame) average over time and the full pool window, 1x1x1 conv + BN + ReLU.

    ``forward`` returns the pooled features at pooled resolution, (B, C', T, H/ph, W/pw) per pool
    size, and ``TemporalBlock`` decides how to merge them.  With the reference's only setting
    (pool = the whole plane, temporal_model.py:24) that is (B, C', T, 1, 1)."""

    def __init__(self, in_channels, reduction_channels, pool_sizes):
        super().__init__()
        feats = []
        self.pool_sizes = [tuple(p) for p in pool_sizes]
        for pool_size in self.pool_sizes:
            assert pool_size[0] == 2, 'time kernel must be 2'
            feats.append(nn.Sequential(OrderedDict([
                ('avgpool', nn.AvgPool3d(kernel_size=pool_size, stride=(1, *pool_size[1:]),
                                         padding=(pool_size[0] - 1, 0, 0), count_include_pad=False)),
                ('conv_bn_relu', conv_1x1x1_norm_activated(in_channels, reduction_channels))])))
        self.features = nn.ModuleList(feats)

    def forward(self, x):
        out = []
        for f, pool in zip(self.features, self.pool_sizes):
            _, ph, pw = pool
            b, c, t, h, w = x.shape
            if h % ph == 0 and w % pw == 0:
                # spatial mean over each pool window, then the causal 2-frame mean with
                # count_include_pad=False (frame 0 averages only itself): identical to the padded
                # AvgPool3d + [:, :, :-1] of the reference (temporal.py:396-413), as plain reductions
                sp = x.float().view(b, c, t, h // ph, ph, w // pw, pw).mean(dim=(4, 6))
                prev = torch.cat([sp[:, :, :1], sp[:, :, :-1]], dim=2)
                pooled = torch.cat([sp[:, :, :1], 0.5 * (sp[:, :, 1:] + prev[:, :, 1:])], dim=2).to(x.dtype)
                out.append(f.conv_bn_relu(pooled))
            else:
                out.append(f(x)[:, :, :-1])
        return out


class TemporalBlock(nn.Module):
    """Three paths (1x1x1 -> causal 2x3x3 ; 1x1x1 -> 1x3x3 ; 1x1x1) + pyramid pooling, concatenated,
    aggregated by a 1x1x1 conv, plus a (projected) skip."""

    def __init__(self, in_channels, out_channels=None, use_pyramid_pooling=False, pool_sizes=None):
        super().__init__()
        self.in_channels = in_channels
        self.half_channels = in_channels // 2
        self.out_channels = out_channels or in_channels
        self.kernels = [(2, 3, 3), (1, 3, 3)]
        self.use_pyramid_pooling = use_pyramid_pooling
        paths = [nn.Sequential(conv_1x1x1_norm_activated(in_channels, self.half_channels),
                               CausalConv3d(self.half_channels, self.half_channels, kernel_size=k))
                 for k in self.kernels]
        paths.append(conv_1x1x1_norm_activated(in_channels, self.half_channels))
        self.convolution_paths = nn.ModuleList(paths)
        agg_in = len(paths) * self.half_channels
        self._paths_channels = agg_in
        if use_pyramid_pooling:
            assert pool_sizes is not None
            reduction = in_channels // 3
            self.pyramid_pooling = PyramidSpatioTemporalPooling(in_channels, reduction, pool_sizes)
            agg_in += len(pool_sizes) * reduction
        self.aggregation = nn.Sequential(conv_1x1x1_norm_activated(agg_in, self.out_channels))
        if self.out_channels != self.in_channels:
            self.projection = nn.Sequential(nn.Conv3d(in_channels, self.out_channels, kernel_size=1, bias=False),
                                            nn.BatchNorm3d(self.out_channels))
        else:
            self.projection = None

    def forward(self, x):
        b, _, t, h, w = x.shape
        paths = torch.cat([p(x) for p in self.convolution_paths], dim=1)
        agg = self.aggregation[0]
        if not self.use_pyramid_pooling:
            y = agg.conv(paths)
        else:
            wgt = agg.conv.weight
            y = F.conv3d(paths, wgt[:, :self._paths_channels])
            off = self._paths_channels
            for pooled in self.pyramid_pooling(x):
                c = pooled.shape[1]
                contrib = F.conv3d(pooled.to(y.dtype), wgt[:, off:off + c])
                if contrib.shape[-2:] != (h, w):
                    if contrib.shape[-2:] == (1, 1):
                        pass                                         # broadcast == bilinear upsample of 1x1
                    else:
                        contrib = F.interpolate(contrib.permute(0, 2, 1, 3, 4).reshape(b * t, -1, *contrib.shape[-2:]),
                                                (h, w), mode='bilinear', align_corners=False)
                        contrib = contrib.view(b, t, -1, h, w).permute(0, 2, 1, 3, 4)
                y = y + contrib
                off += c
        y = agg.activation(agg.norm(y))
        skip = x if self.projection is None else self.projection(x)
        return skip + y
