"""BatchNorm + activation (+ residual) as ONE operator.

The reference chains ``nn.BatchNorm2d/3d -> nn.ReLU`` (or swish in the EfficientNet trunk) and adds
skips as separate elementwise passes (stp3/layers/convolutions.py:183-280, stp3/layers/temporal.py:
252-273, 315-325, 426-489, stp3/models/decoder.py:22-140).  ``bn_act`` takes the BatchNorm *module*
(so parameter / buffer names, momentum, eps and train/eval state stay exactly the reference's) and
runs the whole chain through the HIP kernels of ``stp3_bnact.hip`` on GPU tensors; with more than one
rank the batch statistics are all-reduced (the reference trains with ``sync_batchnorm=True``,
train.py:47).

CPU tensors (the gloo data-parallel tests, the CPU port timed by ``bench.py``) take ``bn_act_reference``: the same
arithmetic written with torch ops.  GPU tensors never fall back: a
missing ``libstp3hip.so`` raises.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..utils import hp

ACT_NONE, ACT_RELU, ACT_SWISH = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_SWISH
RES_NONE, RES_BEFORE_ACT, RES_AFTER_ACT = ops.RES_NONE, ops.RES_BEFORE_ACT, ops.RES_AFTER_ACT


def _sync_world(bn):
    if not (bn.training and dist.is_available() and dist.is_initialized()):
        return 1
    if getattr(bn, 'stp3_local_stats', False):
        return 1
    return dist.get_world_size()


def _act(act, y):
    if act == ACT_RELU:
        return F.relu(y)
    if act == ACT_SWISH:
        return F.silu(y)
    return y


def bn_act_reference(bn, x, act=ACT_NONE, res=None, res_mode=RES_NONE, sbias=None, oscale=None):
    """Plain-torch statement of ``bn_act`` (any device, any rank count; differentiable)."""
    c = bn.num_features
    if x.shape[1] != c:                             # zero-padded channel lanes (see ``bn_act``)
        y = bn_act_reference(bn, x[:, :c], act, None if res is None else res[:, :c], res_mode, sbias, oscale)
        return F.pad(y, [0, 0] * (x.dim() - 2) + [0, x.shape[1] - c])
    shape = [1, -1] + [1] * (x.dim() - 2)
    lead = [x.shape[0], -1] + [1] * (x.dim() - 2)
    if sbias is not None:
        x = x + sbias.to(x.dtype).view(lead)
    use_batch = bn.training or not bn.track_running_stats
    if use_batch:
        dims = [0] + list(range(2, x.dim()))
        xf = hp(x)
        count = float(xf.numel() // xf.shape[1])
        s, q = xf.sum(dims), (xf * xf).sum(dims)
        if _sync_world(bn) > 1:
            import torch.distributed.nn.functional as dfn
            packed = dfn.all_reduce(torch.cat([s, q]))
            s, q = packed[:s.numel()], packed[s.numel():]
            count *= dist.get_world_size()
        mean = s / count
        var = (q / count - mean * mean).clamp_min(0.0)
        if bn.training and bn.track_running_stats:
            with torch.no_grad():
                mom = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (count / max(count - 1.0, 1.0)), alpha=mom)
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked.add_(1)
    else:
        mean, var = hp(bn.running_mean), hp(bn.running_var)
    scale = torch.rsqrt(var + bn.eps)
    shift = -mean * scale
    if bn.weight is not None:
        scale = scale * hp(bn.weight)
        shift = shift * hp(bn.weight) + hp(bn.bias)
    y = (hp(x) * scale.view(shape) + shift.view(shape)).to(x.dtype)
    if res is not None and res_mode == RES_BEFORE_ACT:
        y = y + res.to(y.dtype)
    y = _act(act, y)
    if oscale is not None:
        y = y * oscale.to(y.dtype).view([-1] + [1] * (y.dim() - 1))
    if res is not None and res_mode == RES_AFTER_ACT:
        y = y + res.to(y.dtype)
    return y


flush_batch_counters = ops.flush_batch_counters          # re-export (parallel.FlatAdam, tests)


def bn_act(bn, x, act=ACT_NONE, res=None, res_mode=RES_NONE, sbias=None, oscale=None):
    """y = act(bn(x + sbias) [+ res if BEFORE_ACT]) * oscale [+ res if AFTER_ACT].

    ``bn``: the nn.BatchNorm2d / nn.BatchNorm3d / nn.SyncBatchNorm module (its forward is not called);
    x (N, C, H, W); sbias (N, C) per-sample bias; oscale (N,) per-sample scale; res like x.

    Zero-padded channel lanes: x (and res) may carry ``pad8(bn.num_features)`` channels, the extra ones padding (the
    output channels a convolution with zero-padded weights produced).  They are ignored on input and zero in the result,
    which keeps the padded shape: a 35-channel layer then lives in 40-lane rows that the MFMA convolutions (channel
    counts in multiples of 8) read and write in place, with no pad / slice copies between the operators."""
    if res is None:
        res_mode = RES_NONE
    if (x.is_cuda and x.dim() == 5 and x.shape[3] == 1 and x.shape[4] == 1 and res is None and sbias is None
            and oscale is None):
        # BatchNorm3d of a (B, C, T, 1, 1) descriptor (the pyramid-pooling branch): per-channel statistics over the
        # B*T vectors, i.e. the 4-D operator on (B*T, C, 1, 1) -- a handful of kernel launches instead of the ~50
        # tiny torch operators of the plain statement and its backward
        b, c, t = x.shape[:3]
        y = bn_act(bn, x.permute(0, 2, 1, 3, 4).reshape(b * t, c, 1, 1), act)
        return y.view(b, t, c, 1, 1).permute(0, 2, 1, 3, 4)
    if not (x.is_cuda and x.dim() == 4) or x.dtype == torch.float64:
        # CPU tensors, and float64 on any device: the library has float32 / bf16 kernels only; a float64 evaluation (the
        # noise-free truth of the parity tests) takes the torch statement
        return bn_act_reference(bn, x, act, res, res_mode, sbias, oscale)
    if torch.is_autocast_enabled() and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype('cuda'))
    training = bn.training or not bn.track_running_stats
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        ops.bump_batch_counter(bn)
    group = None if _sync_world(bn) > 1 else False
    return ops.bn_act(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                      bn.running_var if bn.track_running_stats else None, training, bn.momentum, bn.eps,
                      act=act, res=res, res_mode=res_mode, sbias=sbias, oscale=oscale, group=group,
                      channels=bn.num_features if x.shape[1] != bn.num_features else None)


# Dense convolutions.  Every bf16 (autocast) convolution on the GPU -- forward, data gradient and weight gradient, all
# layer shapes of the model, the 3-channel stem included (its channels are zero-padded to 8) -- runs on the hand-written
# MFMA implicit-GEMM kernels of stp3_conv.hip; no vendor convolution is on the benchmarked path.  Float32 tensors outside
# autocast (the float32 parity runs) and CPU tensors take torch's convolution.
def _use_mfma(x, weight, stride):
    if not x.is_cuda:
        return False
    if not (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()):
        return False
    return ops.conv2d_supported(x, weight, stride)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    if _use_mfma(x, weight, stride):
        return ops.conv2d(x, weight, bias, stride, padding, dilation)
    return F.conv2d(x, weight, bias, stride, padding, dilation)


class _PlaneMean(torch.autograd.Function):
    """Mean over the plane of every (sample, channel) of an (N, C, H, W) tensor, float32 result (N, C).  torch's own
    ``mean`` / ``AdaptiveAvgPool2d(1)`` give the same value, but their backward hands back the broadcast gradient in
    NCHW-contiguous memory whatever the layout of x; for a channels-last x that puts every later addition of input
    gradients on torch's strided element-wise kernel (3x slower).  Here the gradient is laid out like x."""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (tuple(x.shape), x.dtype, x.is_contiguous(memory_format=torch.channels_last))
        return x.mean(dim=(2, 3), dtype=torch.float64 if x.dtype == torch.float64 else torch.float32)

    @staticmethod
    def backward(ctx, g):
        (n, c, h, w), dtype, cl = ctx.meta
        g = (g * (1.0 / (h * w))).to(dtype).view(n, c, 1, 1).expand(n, c, h, w)
        return g.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)


def plane_mean(x):
    """(N, C, H, W) -> (N, C) float32 mean over H x W (see ``_PlaneMean``)."""
    return _PlaneMean.apply(x)


def conv1x1_on_vector(x, weight, bias=None):
    """A 1x1 convolution of a (N, C, 1, 1) map (or (N, C, T, 1, 1) with a 1x1x1 kernel) is a plain GEMM."""
    w2 = weight.flatten(1)
    if x.dim() == 5:
        n, c, t = x.shape[:3]
        y = F.linear(x.reshape(n, c, t).transpose(1, 2), w2, bias)           # (N, T, Co)
        return y.transpose(1, 2).reshape(n, -1, t, 1, 1)
    y = F.linear(x.flatten(1), w2, bias)
    return y.view(*y.shape, 1, 1)


def conv_module(m, x):
    """``m(x)`` for an ``nn.Conv2d`` through ``conv2d`` when it is a plain dense zero-padded convolution."""
    if m.groups == 1 and m.padding_mode == 'zeros' and not isinstance(m.padding, str):
        if m.kernel_size == (1, 1) and x.shape[-2:] == (1, 1):
            return conv1x1_on_vector(x, m.weight, m.bias)
        return conv2d(x, m.weight, m.bias, m.stride, m.padding, m.dilation)
    return m(x)


def _fusable_conv_bn(conv, bn, x):
    """conv -> BN (-> ReLU) in training mode goes through ONE operator whose convolution epilogue produces the
    BatchNorm statistics (ops_fused.conv_bn_act): the statistics pass over the convolution output disappears."""
    return (type(conv) is nn.Conv2d and isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training
            and bn.track_running_stats and conv.groups == 1 and conv.padding_mode == 'zeros'
            and not isinstance(conv.padding, str) and x.dim() == 4 and _use_mfma(x, conv.weight, conv.stride)
            and ops.conv2d_stats_supported(x, conv.weight, conv.stride, conv.padding, conv.dilation))


def run_fused(seq, x):
    """Run an ``nn.Sequential`` with every ``BatchNorm -> ReLU`` pair (or lone BatchNorm) fused."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if i + 1 < len(mods) and _fusable_conv_bn(m, mods[i + 1], x):
            from .. import ops_fused
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            group = None if _sync_world(mods[i + 1]) > 1 else False
            x = ops_fused.conv_bn_act(x, m.weight, m.bias, mods[i + 1], ACT_RELU if relu else ACT_NONE, None, RES_NONE,
                                      m.stride, m.padding, m.dilation, group=group)
            i += 3 if relu else 2
            continue
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                x = bn_act(m, x, ACT_RELU)
                i += 2
                continue
            x = bn_act(m, x, ACT_NONE)
        elif isinstance(m, nn.Sequential):
            x = run_fused(m, x)
        elif type(m) is nn.Conv2d:
            x = conv_module(m, x)
        else:
            x = m(x)
        i += 1
    return x
