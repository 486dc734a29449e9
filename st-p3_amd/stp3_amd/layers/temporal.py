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

from .. import ops
from ..utils import hp
from .fused import (ACT_NONE, ACT_RELU, RES_AFTER_ACT, bn_act, bn_act_group, conv1x1_on_vector, conv2d, plane_mean, pooled_bias,
                    slot_ok)


def _pad8(c):
    return (c + 7) // 8 * 8


# Zero-padded channel lanes.  The first TemporalBlock works on 35-channel tensors (half of 64 + 6); the MFMA convolutions
# take channel counts in multiples of 8.  On the GPU those layers therefore live in 40-LANE rows from end to end: the
# (small) weights get zero rows / columns, the convolutions produce exact zeros in lanes 35..39, the fused BatchNorm
# (``fused.bn_act``: stp3_bn_dims.cpad) ignores them on input and writes zeros there, and nothing pads, slices or
# copies an activation (1.2 ms per step of torch pad / slice_backward / contiguous copies before).
def _pad_out(w, lanes):
    """(Co, Ci, kh, kw) weight with its output channels zero-padded to ``lanes``."""
    co = w.shape[0]
    return w if co == lanes else F.pad(w, (0, 0, 0, 0, 0, 0, 0, lanes - co))


def _pad_in(w, groups, lanes):
    """(Co, groups * g, kh, kw) weight whose input channels are ``groups`` runs of g, each run zero-padded to ``lanes``
    (the layout of a concatenation of ``groups`` lane-padded tensors)."""
    co, ci, kh, kw = w.shape
    g = ci // groups
    if g == lanes:
        return w
    return F.pad(w.reshape(co, groups, g, kh, kw), (0, 0, 0, 0, 0, lanes - g)).reshape(co, groups * lanes, kh, kw)


def _conv2d_padded_channels(x, weight, padding=0):
    """conv2d with input/output channels zero-padded to multiples of 8 (exact: the extra channels carry
    zeros).  The tuned bf16 channels-last convolution kernels on MI355X need 16-byte channel vectors; the
    TemporalBlock's 35 / 70 / 23 / 117-channel layers otherwise fall onto generic kernels that are
    ~100x slower (profiles/r01_*)."""
    co, ci = weight.shape[:2]
    cop, cip = _pad8(co), _pad8(ci)
    if cip != ci:
        x = F.pad(x, (0, 0, 0, 0, 0, cip - ci))
        weight = F.pad(weight, (0, 0, 0, 0, 0, cip - ci))
    if cop != co:
        weight = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, cop - co))
    y = conv2d(x, weight, None, 1, padding)
    return y[:, :co] if cop != co else y


def _conv2d_pieces(x, key, out_channels, kernel, pieces, padding=0, direct=True, token_out=None):
    """conv2d of x with the weight ASSEMBLED from ``pieces`` (``ops.weight_piece``: views of parameters at channel offsets of a
    zero weight with ``out_channels`` outputs -- padded to a multiple of 8 -- and x's channels as inputs): what
    ``_conv2d_padded_channels(x, <the same weight built with pad / cat>)`` computes, without a torch operator for the weight
    or its gradient (``ops.assembled_weight``).  Callers check ``ops.assembled_weight_supported`` and x.shape[1] % 8 == 0."""
    cop = _pad8(out_channels)
    w = ops.assembled_weight(key, (cop, x.shape[1], kernel[0], kernel[1]), pieces, direct=direct)
    if token_out is not None:
        token_out.append(w)                        # (for ``ops.weight_columns``: the parameter's other columns)
    y = conv2d(x, w, None, 1, padding)
    return y[:, :out_channels] if cop != out_channels else y


def _bn_act_2d(norm, x, relu=True, res=None, sbias=None, out_slot=None):
    """BatchNorm3d over (B,C,T,H,W) == batch norm over (B*T,C,H,W): apply the 3-D module's statistics
    and affine parameters to the frame-folded 4-D tensor (fused with the ReLU / skip add)."""
    return bn_act(norm, x, ACT_RELU if relu else ACT_NONE, res=res, res_mode=RES_AFTER_ACT, sbias=sbias, out_slot=out_slot)


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

    def forward_folded(self, x2, batch, frames, out_slot=None):
        """Same layer on the frame-folded tensor x2 (B*T, C, H, W).  A causal (2,3,3) convolution is
        y[t] = W[:, :, 0] * x[t-1] + W[:, :, 1] * x[t] with x[-1] = 0, i.e. ONE 2-D 3x3 convolution over
        the channel pairing [x[t-1], x[t]] (one kernel pass: ``ops.causal_pair``); a (1,3,3) convolution is a per-frame
        2-D 3x3.  x2 may carry zero-padded channel lanes (see ``_pad_out``); on the GPU the result does too."""
        w = self.conv.weight
        kt = w.shape[2]
        assert self.conv.bias is None and self.conv.dilation == (1, 1, 1) and kt in (1, 2)
        lanes_in = x2.shape[1]
        if lanes_in % 8 == 0 and ops.assembled_weight_supported(x2, (w,)) and (kt == 1 or ops.causal_pair_supported(x2)):
            # the taps side by side, each in its own run of ``lanes_in`` input lanes, output lanes padded to 8: pieces of one
            # assembled weight (no unbind / pad / cat, and none of their backward)
            taps = w.detach().unbind(2)
            pieces = [ops.weight_piece(w, taps[k], 0, k * lanes_in) for k in range(kt)]
            if kt == 2:
                x2 = ops.causal_pair(x2, frames)
            y = _conv2d_pieces(x2, (id(self), 'folded'), _pad8(w.shape[0]), w.shape[3:], pieces, padding=self._hw_pad[1:])
            return _bn_act_2d(self.norm, y, out_slot=out_slot)
        # (unbind / squeeze, not w[:, :, k]: the backward of k selects is k zero-fills, k copies and k - 1 additions; that of
        # an unbind one stack, that of a squeeze nothing)
        taps = [_pad_in(wk, 1, lanes_in) for wk in (w.unbind(2) if kt > 1 else (w.squeeze(2),))]
        if kt == 2:
            if ops.causal_pair_supported(x2):
                x2 = ops.causal_pair(x2, frames)
            else:
                c, h, ww = x2.shape[1:]
                x5 = x2.view(batch, frames, c, h, ww)
                prev = torch.cat([torch.zeros_like(x5[:, :1]), x5[:, :-1]], dim=1).view(batch * frames, c, h, ww)
                x2 = torch.cat([prev, x2], dim=1)
            w2 = torch.cat(taps, dim=1)
        else:
            w2 = taps[0]
        if x2.is_cuda:
            w2 = _pad_out(w2, _pad8(w2.shape[0]))
        y = _conv2d_padded_channels(x2, w2, padding=self._hw_pad[1:])
        return _bn_act_2d(self.norm, y, out_slot=out_slot)


def conv_1x1x1_norm_activated(in_channels, out_channels):
    return nn.Sequential(OrderedDict([('conv', nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False)),
                                      ('norm', nn.BatchNorm3d(out_channels)),
                                      ('activation', nn.ReLU(inplace=True))]))


class Bottleneck3D(nn.Module):
    """1x1x1 down-projection -> causal (kt, 3, 3) convolution -> 1x1x1 up-projection, plus the (projected) skip
    (temporal.py:328-375).  Only built for ``MODEL.TEMPORAL_MODEL.INBETWEEN_LAYERS > 0``, which no shipped configuration sets:
    the layers run as the plain torch modules they are (off the benchmarked path; parameter names as in the reference)."""

    def __init__(self, in_channels, out_channels=None, kernel_size=(2, 3, 3), dilation=(1, 1, 1)):
        super().__init__()
        half = in_channels // 2
        out_channels = out_channels or in_channels
        self.layers = nn.Sequential(OrderedDict([
            ('conv_down_project', conv_1x1x1_norm_activated(in_channels, half)),
            ('conv', CausalConv3d(half, half, kernel_size=kernel_size, dilation=dilation, bias=False)),
            ('conv_up_project', conv_1x1x1_norm_activated(half, out_channels))]))
        self.projection = None
        if out_channels != in_channels:
            self.projection = nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size=1, bias=False),
                                            nn.BatchNorm3d(out_channels))

    def forward(self, x):
        skip = x if self.projection is None else self.projection(x)
        return self.layers(x) + skip


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

    def whole_plane_members(self, x, extra=None, folded=None):
        """The whole-plane pooling branches (the reference's only setting) as members of a ``fused.bn_act_group`` plus the
        function that turns each member's output into ``forward``'s (B, C', T, 1, 1) -- or None when a branch pools
        smaller windows (then ``forward`` is the way).  Their BatchNorms are siblings of the pointwise convolutions at the
        head of the temporal block: one statistics exchange for all of them."""
        b, c, t, h, w = x.shape
        if folded is None or any((ph, pw) != (h, w) for _, ph, pw in self.pool_sizes):
            return None
        members, finish = [], []
        for f in self.features:
            sp = plane_mean(folded).view(b, t, c).permute(0, 2, 1)[..., None, None]
            if extra is not None:
                sp = torch.cat([sp, hp(extra).view(b, -1, t, 1, 1)], dim=1)
            pooled = torch.cat([sp[:, :, :1], 0.5 * (sp[:, :, 1:] + sp[:, :, :-1]), sp[:, :, -1:]], dim=2)
            cbr = f.conv_bn_relu
            y = conv1x1_on_vector(pooled, cbr.conv.weight)                       # (B, C', T + 1, 1, 1)
            co = y.shape[1]
            members.append(dict(bn=cbr.norm, x=y.permute(0, 2, 1, 3, 4).reshape(b * (t + 1), co, 1, 1), act=ACT_RELU))
            finish.append(lambda o, co=co: o.view(b, t + 1, co, 1, 1).permute(0, 2, 1, 3, 4)[:, :, :-1])
        return members, finish

    def forward(self, x, extra=None, folded=None):
        """``extra`` (B, E, T): channels that are constant over each frame's plane (the ego-motion planes of
        stp3.py:145-152) -- their window mean is the value itself, so they join after the spatial mean.
        ``folded``: the same input as a frame-folded (B*T, C, H, W) tensor (what ``TemporalBlock`` holds anyway, in
        channels-last memory): whole-plane means are taken from it -- a column reduction over contiguous rows instead
        of a strided 7-D reduction, and its gradient joins the block's other input gradients in the same layout."""
        out = []
        for f, pool in zip(self.features, self.pool_sizes):
            _, ph, pw = pool
            b, c, t, h, w = x.shape
            if extra is not None:
                assert h == ph and w == pw, 'constant planes are folded for whole-plane pooling only'
            if folded is not None and h == ph and w == pw:
                sp = plane_mean(folded).view(b, t, c).permute(0, 2, 1)[..., None, None]
                if extra is not None:
                    sp = torch.cat([sp, hp(extra).view(b, -1, t, 1, 1)], dim=1)
                pooled = torch.cat([sp[:, :, :1], 0.5 * (sp[:, :, 1:] + sp[:, :, :-1]), sp[:, :, -1:]], dim=2)
                cbr = f.conv_bn_relu
                out.append(bn_act(cbr.norm, conv1x1_on_vector(pooled, cbr.conv.weight), ACT_RELU)[:, :, :-1])
            elif h % ph == 0 and w % pw == 0:
                # spatial mean over each pool window, then the causal 2-frame mean with
                # count_include_pad=False (frame 0 averages only itself): identical to the padded
                # AvgPool3d + [:, :, :-1] of the reference (temporal.py:396-413), as plain reductions
                sp = hp(x).view(b, c, t, h // ph, ph, w // pw, pw).mean(dim=(4, 6))
                if extra is not None:
                    sp = torch.cat([sp, hp(extra).view(b, -1, t, 1, 1)], dim=1)
                # T+1 causal windows: {0}, {0,1}, ..., {T-2,T-1}, {T-1}.  The reference runs conv+BN+ReLU on
                # all T+1 (so the BatchNorm batch statistics include the last, right-padded window) and
                # only then drops it.
                pooled = torch.cat([sp[:, :, :1], 0.5 * (sp[:, :, 1:] + sp[:, :, :-1]), sp[:, :, -1:]], dim=2)
                cbr = f.conv_bn_relu
                out.append(bn_act(cbr.norm, conv1x1_on_vector(pooled, cbr.conv.weight), ACT_RELU)[:, :, :-1])
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

    @staticmethod
    def _pointwise_member(seq, x2, relu=True, extra2=None, lanes=None):
        """conv_1x1x1_norm_activated (or projection) on the frame-folded tensor: a 1x1 2-D convolution, returned as the
        arguments of its BatchNorm (``fused.bn_act`` / ``bn_act_group``).
        ``extra2`` (B*T, E): input channels that are constant over the plane -- their part of the 1x1 convolution is
        a per-frame bias, W[:, C:] @ extra, added inside the fused BatchNorm (exact; no concatenated tensor).
        ``lanes``: output channel lanes (>= the layer's channels; the extra ones come out zero, see ``_pad_out``)."""
        conv, norm = seq[0], seq[1]
        wgt = conv.weight.squeeze(2)                                     # (1x1x1 kernel: a view, nothing to add up in backward)
        c = x2.shape[1]
        if c % 8 == 0 and ops.assembled_weight_supported(x2, (conv.weight,)):
            # the columns that multiply x2, output lanes padded: one piece of an assembled weight.  With ``extra2`` the
            # parameter's other columns take their own way to the loss (below): its gradient is put together by autograd.
            piece = ops.weight_piece(conv.weight, conv.weight.detach().squeeze(2)[:, :c])
            token = []
            y = _conv2d_pieces(x2, (id(conv), 'x'), wgt.shape[0] if lanes is None else lanes, (1, 1), [piece],
                               direct=True if extra2 is None else 'shared', token_out=token)
            w_extra = None if extra2 is None else ops.weight_columns(token[0], conv.weight, c, wgt.shape[1])
        else:
            # (split, not two slices: one concatenation in backward instead of two zero-fills, two copies and an addition)
            w_x, w_extra = (wgt, None) if extra2 is None else wgt.split([c, wgt.shape[1] - c], dim=1)
            y = _conv2d_padded_channels(x2, w_x if lanes is None else _pad_out(w_x, lanes))
        # (through conv1x1_on_vector: float32 with autocast off on the GPU -- a (B*T, 6) x (6, C') product needs no casts)
        sbias = None if extra2 is None else conv1x1_on_vector(hp(extra2).to(hp(wgt).dtype)[:, :, None, None], w_extra).flatten(1)
        return dict(bn=norm, x=y, act=ACT_RELU if relu else ACT_NONE, sbias=sbias)

    @staticmethod
    def _pointwise(seq, x2, relu=True, extra2=None, lanes=None):
        m = TemporalBlock._pointwise_member(seq, x2, relu, extra2, lanes)
        return _bn_act_2d(m['bn'], m['x'], relu, sbias=m['sbias'])

    def forward(self, x, extra=None):
        """x (B, C, T, H, W) -> (B, C', T, H, W).  Everything runs frame-folded as 2-D ops on
        (B*T, C, H, W): the block only couples frames through the causal 2-tap convolution and the
        causal pyramid pooling, both of which are expressed explicitly.  ``extra`` (B, E, T): the last E of the
        block's ``in_channels`` given as per-frame constants instead of planes (every consumer of the block's input is
        a 1x1x1 convolution or the whole-plane pooling, so the fold is exact)."""
        b, c, t, h, w = x.shape
        extra2 = None if extra is None else extra.permute(0, 2, 1).reshape(b * t, -1)
        assert c + (0 if extra is None else extra.shape[1]) == self.in_channels
        x2 = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        if x2.is_cuda:
            x2 = x2.contiguous(memory_format=torch.channels_last)
            if torch.is_autocast_enabled() and x2.dtype == torch.float32:
                # the float32 BEV feeds five consumers of this block: cast it ONCE (each convolution would otherwise
                # cast its own copy, and the five float32 input gradients would be added up in float32)
                x2 = x2.to(torch.get_autocast_dtype('cuda'))
        # the half_channels-wide paths run in lanes of 8 channels on the GPU (see ``_pad_out``)
        lanes = _pad8(self.half_channels) if x2.is_cuda else self.half_channels
        # the pointwise convolutions at the head of the block (two paths, the third path, the skip projection) are
        # siblings: one BatchNorm statistics exchange for all of them when the statistics are shared between ranks
        # x2 feeds the pointwise convolutions, the whole-plane pooling and (without a projection) the skip: their input
        # gradients are added in one pass (ops.fan_out) instead of pairwise
        n_use = len(self.convolution_paths) + 1 + int(self.use_pyramid_pooling)
        xs = ops.fan_out(x2, n_use) if x2.is_cuda else [x2] * n_use
        heads = [self._pointwise_member(path[0], xi, extra2=extra2, lanes=lanes)
                 for path, xi in zip(self.convolution_paths[:-1], xs)]
        k = len(heads)
        heads.append(self._pointwise_member(self.convolution_paths[-1], xs[k], extra2=extra2, lanes=lanes))
        # The three path outputs are WRITTEN into the channel slices of one buffer (the aggregation's operand) instead of being
        # concatenated: two concatenations of 3 x (12, 40 | 32, 200, 200) per step, 98 + 76 us (profiles/r05last_step_trace.txt)
        joined = None
        if slot_ok(heads[k]['x']) and heads[k]['x'].shape[1] == lanes:
            from .. import ops_fused
            joined = torch.empty((b * t, (k + 1) * lanes, h, w), dtype=heads[k]['x'].dtype, device=x2.device,
                                 memory_format=torch.channels_last)
            heads[k]['out_slot'] = (joined, k * lanes)
        if self.projection is not None:
            heads.append(self._pointwise_member(self.projection, xs[k + 1], relu=False, extra2=extra2))
        x_skip = xs[k + 1]                                               # (only used when there is no projection)
        n_pointwise = len(heads)
        x_pool = xs[k + 2] if self.use_pyramid_pooling else x2
        pooled_group = self.pyramid_pooling.whole_plane_members(x, extra, folded=x_pool) if self.use_pyramid_pooling else None
        if pooled_group is not None:
            heads += pooled_group[0]
        heads = bn_act_group(heads)
        pooled_outs = None if pooled_group is None else [fin(o) for fin, o in zip(pooled_group[1], heads[n_pointwise:])]
        heads = heads[:n_pointwise]
        outs = [path[1].forward_folded(y, b, t, out_slot=None if joined is None else (joined, i * lanes))
                for i, (path, y) in enumerate(zip(self.convolution_paths[:-1], heads))]
        outs.append(heads[len(self.convolution_paths) - 1])
        paths = torch.cat(outs, dim=1) if joined is None else ops_fused.join_slices(joined, outs)
        agg = self.aggregation[0]
        wgt = agg.conv.weight.squeeze(2)                                 # (Cout, Cin_total, 1, 1)
        # the paths' columns and one run of columns per pooled tensor: ONE split (its backward is one concatenation)
        pooled_list = [] if not self.use_pyramid_pooling else \
            list(pooled_outs if pooled_outs is not None else self.pyramid_pooling(x, extra, folded=x_pool))   # (B, C', T, h', w')
        if lanes % 8 == 0 and paths.shape[1] == len(outs) * lanes and ops.assembled_weight_supported(paths, (agg.conv.weight,)):
            # one run of columns per path, each at the start of its run of ``lanes`` input lanes: pieces of an assembled weight;
            # the pooled tensors' columns go their own way (below) and write their part of the gradient themselves
            g = self._paths_channels // len(outs)
            flat = agg.conv.weight.detach().squeeze(2)
            pieces = [ops.weight_piece(agg.conv.weight, flat[:, i * g:(i + 1) * g], 0, i * lanes) for i in range(len(outs))]
            token = []
            y = _conv2d_pieces(paths, (id(agg.conv), 'paths'), wgt.shape[0], (1, 1), pieces,
                               direct='shared' if pooled_list else True, token_out=token)
            w_parts, c0 = [None], self._paths_channels
            for pl in pooled_list:
                w_parts.append(ops.weight_columns(token[0], agg.conv.weight, c0, c0 + pl.shape[1]))
                c0 += pl.shape[1]
        else:
            w_parts = wgt.split([self._paths_channels] + [pl.shape[1] for pl in pooled_list], dim=1) if pooled_list else (wgt,)
            y = _conv2d_padded_channels(paths, _pad_in(w_parts[0], len(outs), lanes))
        sbias = None
        if self.use_pyramid_pooling:
            for pooled, w_p in zip(pooled_list, w_parts[1:]):
                cp = pooled.shape[1]
                p2 = pooled.permute(0, 2, 1, 3, 4).reshape(b * t, cp, *pooled.shape[-2:])
                contrib = (conv1x1_on_vector(p2, w_p)
                           if p2.shape[-2:] == (1, 1) else F.conv2d(p2.to(y.dtype), w_p.to(y.dtype)))
                if contrib.shape[-2:] == (1, 1):
                    # whole-plane pooling (the reference's only setting): a constant plane per frame, i.e. a
                    # per-sample bias of the aggregation -- folded into the fused BatchNorm
                    contrib = hp(contrib.flatten(1))
                    sbias = contrib if sbias is None else sbias + contrib
                else:
                    if contrib.shape[-2:] != (h, w):
                        contrib = F.interpolate(contrib, (h, w), mode='bilinear', align_corners=False)
                    y = y + contrib
        assert self.projection is not None or extra is None
        skip = x_skip if self.projection is None else heads[-1]
        out = _bn_act_2d(agg.norm, y, res=skip, sbias=pooled_bias(sbias))
        return out.view(b, t, -1, h, w).permute(0, 2, 1, 3, 4)


# ----------------------------------------------------------------------------------------------
# Convolutional GRUs of the prediction stage: stp3/layers/temporal.py:11-145.
# ----------------------------------------------------------------------------------------------
def _conv(m, x):
    from .fused import conv_module
    return conv_module(m, x)


# Sequences of the prediction stage as FRAME-MAJOR channels-last memory.  The reference keeps (B,S,C,H,W) tensors and takes
# frames with x[:, t] / builds them with torch.stack(..., dim=1) (temporal.py:34-40, :98-117; future_prediction.py:33-45).  In
# that memory order a frame is neither a dense channels-last tensor (the kernels' layout: a copy per frame and use) nor is its
# gradient (SelectBackward: a zero fill, a copy and an addition per frame).  Here a sequence is stored [S][B][H][W][C] and
# presented as the (B,S,C,H,W) permutation of it: every frame x[:, t] is a dense channels-last (B,C,H,W) view, the whole
# sequence is ONE channels-last batch of S*B frames (frames_as_batch: a view), and the two autograd functions below keep the
# gradients in the same order.  Values and the (B,S,...) indexing are the reference's; only strides differ.
class _StackFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *frames):
        buf = torch.stack([f.permute(0, 2, 3, 1) for f in frames], dim=0)          # [S][B][H][W][C], one launch
        return buf.permute(1, 0, 4, 2, 3)

    @staticmethod
    def backward(ctx, g):
        return tuple(g[:, t] for t in range(g.shape[1]))


class _UnbindFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        ctx.meta = (x.shape, x.dtype, x.device)
        return tuple(x[:, t] for t in range(x.shape[1]))

    @staticmethod
    def backward(ctx, *grads):
        shape, dtype, device = ctx.meta
        b, s, c, h, w = shape
        if all(g is None for g in grads):
            return None
        if all(g is not None for g in grads):
            buf = torch.stack([g.permute(0, 2, 3, 1) for g in grads], dim=0)       # one launch
        else:
            buf = torch.empty((s, b, h, w, c), dtype=dtype, device=device)
            for t, g in enumerate(grads):
                if g is None:
                    buf[t].zero_()
                else:
                    buf[t].copy_(g.permute(0, 2, 3, 1))
        return buf.permute(1, 0, 4, 2, 3)


def stack_frames(frames):
    """list of S (B,C,H,W) tensors -> (B,S,C,H,W) in frame-major channels-last memory (see above)."""
    return _StackFrames.apply(*frames)


def unbind_frames(x):
    """(B,S,C,H,W) -> its S frames; the gradient comes back frame-major."""
    return _UnbindFrames.apply(x)


def frames_as_batch(x):
    """(B,S,C,H,W) -> (a (S*B,C,H,W) channels-last batch, restore): per-frame modules (convolutions, BatchNorm over all
    frames, LayerNorm) do not care about the order of the frames in the batch, so a frame-major sequence goes in as the VIEW
    it already is -- frame-major batch order -- and ``restore`` turns the result back into a (B,S,C',H',W') sequence.
    Anything else is flattened like the reference does (``x.view(b * s, c, h, w)``)."""
    b, s, c, h, w = x.shape
    fm = x.permute(1, 0, 3, 4, 2)
    if s > 1 and fm.is_contiguous():
        def restore(y):
            y = y.permute(0, 2, 3, 1)
            if not y.is_contiguous():
                y = y.contiguous()
            return y.view(s, b, *y.shape[1:]).permute(1, 0, 4, 2, 3)
        return fm.reshape(s * b, h, w, c).permute(0, 3, 1, 2), restore
    return x.reshape(b * s, c, h, w), lambda y: y.view(b, s, *y.shape[1:])


def batch_major(x):
    """A frame-major sequence as (B,S,C,H,W) with [B][S][H][W][C] memory: what ``x.view(b * s, c, h, w)`` of the consumers
    behind the prediction stage (the decoder) flattens without a copy into a channels-last batch.  One transposing copy."""
    b, s, c, h, w = x.shape
    if s == 1 or not x.permute(1, 0, 3, 4, 2).is_contiguous():
        return x
    out = torch.empty((b, s, h, w, c), dtype=x.dtype, device=x.device)
    out.copy_(x.permute(0, 1, 3, 4, 2))
    return out.permute(0, 1, 4, 2, 3)


def _gru_cell(x, state, conv_update, conv_reset, conv_state_tilde, bias_init):
    """One convolutional GRU step (temporal.py:42-56): the update and reset gates read the same [x, state] operand, so
    their two 3x3 convolutions run as ONE convolution with the output channels concatenated (exact: the operand tile
    is staged once instead of twice)."""
    from .. import ops_pred
    if ops_pred.gru_cell_supported(x, state, conv_update, conv_reset, conv_state_tilde):
        # bf16 on the GPU: the three convolutions and the gate arithmetic as one operator (two element-wise launches forward,
        # two backward, instead of a dozen and two dozen torch ones: stp3_gru_*); the state stays bf16 along the recurrence
        return ops_pred.gru_cell(x, state, conv_update, conv_reset, conv_state_tilde, bias_init)
    xs = torch.cat([x, state], dim=1)
    hidden = conv_update.out_channels
    w = torch.cat([conv_update.weight, conv_reset.weight], dim=0)
    b = torch.cat([conv_update.bias, conv_reset.bias], dim=0)
    from .fused import conv2d
    gates = torch.sigmoid(hp(conv2d(xs, w, b, 1, conv_update.padding, 1)) + bias_init)
    update, reset = gates.split(hidden, dim=1)
    state_f = hp(state)
    tilde = hp(_conv(conv_state_tilde, torch.cat([x, ((1.0 - reset) * state_f).to(x.dtype)], dim=1)))
    return ((1.0 - update) * state_f + update * tilde).to(state.dtype)


class SpatialGRU(nn.Module):
    """Convolutional GRU over a (B,T,C,H,W) sequence with a 1x1 decoder per step (temporal.py:11-56)."""

    def __init__(self, input_size, hidden_size, gru_bias_init=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        self.conv_update = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_reset = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_state_tilde = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size=3, bias=True, padding=1)
        self.conv_decoder = nn.Conv2d(hidden_size, input_size, kernel_size=1, bias=False)

    def forward(self, x, state=None):
        assert x.dim() == 5, 'Input tensor must be BxTxCxHxW.'
        b, steps, c, h, w = x.shape
        rnn_state = x.new_zeros(b, self.hidden_size, h, w) if state is None else state
        out = []
        for frame in unbind_frames(x):
            rnn_state = self.gru_cell(frame, rnn_state)
            out.append(_conv(self.conv_decoder, rnn_state))
        return stack_frames(out)

    def gru_cell(self, x, state):
        return _gru_cell(x, state.to(x.dtype), self.conv_update, self.conv_reset, self.conv_state_tilde, self.gru_bias_init)


class Dual_GRU(nn.Module):
    """Two coupled convolutional GRUs -- one driven by the latent sample, one by the past states -- mixed by a learned
    per-pixel trust gate (temporal.py:58-145)."""

    def __init__(self, in_channels, latent_dim, n_future, mixture=True, gru_bias_init=0.0):
        super().__init__()
        from .convolutions import Bottleblock
        input_size, hidden_size = in_channels, latent_dim
        self.n_future, self.mixture = n_future, mixture
        self.input_size, self.hidden_size, self.gru_bias_init = input_size, hidden_size, gru_bias_init
        conv3 = lambda ci, co: nn.Conv2d(ci, co, kernel_size=3, bias=True, padding=1)
        self.conv_update_1 = conv3(input_size + hidden_size, hidden_size)
        self.conv_reset_1 = conv3(input_size + hidden_size, hidden_size)
        self.conv_state_tilde_1 = conv3(input_size + hidden_size, hidden_size)
        self.conv_update_2 = conv3(2 * hidden_size, hidden_size)
        self.conv_reset_2 = conv3(2 * hidden_size, hidden_size)
        self.conv_state_tilde_2 = conv3(2 * hidden_size, hidden_size)
        self.conv_decoder_2 = conv3(hidden_size, hidden_size)
        self.trusting_gate = nn.Sequential(Bottleblock(2 * hidden_size, hidden_size),
                                           nn.Conv2d(hidden_size, 2, kernel_size=1, bias=False))

    def forward(self, x, state):
        """x (B,1,input_size,H,W): the latent sample; state (B,n_present,hidden_size,H,W) -> (B,n_future,hidden,H,W)."""
        b, s, c, hh, ww = x.shape
        assert c == self.input_size, f'feature sizes must match, got input {c} for layer with size {self.input_size}'
        n_present = state.shape[1]
        past = unbind_frames(state)
        h = past[0]
        for t in range(n_present - 1):                       # warm-up on the past frames
            h = self.gru_cell_2(past[t], h)
        rnn_state1 = rnn_state2 = past[-1]
        x = x[:, 0]
        pred = []
        for _ in range(self.n_future):
            rnn_state1 = self.gru_cell_1(x, rnn_state1)
            h = self.gru_cell_2(rnn_state2, h)
            rnn_state2 = _conv(self.conv_decoder_2, h)
            mix = torch.cat([rnn_state1, rnn_state2.to(rnn_state1.dtype)], dim=1)
            gate = _conv(self.trusting_gate[1], self.trusting_gate[0](mix))
            gate = torch.softmax(hp(gate), dim=1)
            cur = (hp(rnn_state2) * gate[:, 0:1] + hp(rnn_state1) * gate[:, 1:]).to(rnn_state1.dtype)
            pred.append(cur)
            if self.mixture:
                rnn_state1 = rnn_state2 = cur
        return stack_frames(pred)

    def gru_cell_1(self, x, state):
        return _gru_cell(x.to(state.dtype), state, self.conv_update_1, self.conv_reset_1, self.conv_state_tilde_1,
                         self.gru_bias_init)

    def gru_cell_2(self, x, state):
        return _gru_cell(x.to(state.dtype), state, self.conv_update_2, self.conv_reset_2, self.conv_state_tilde_2,
                         self.gru_bias_init)
