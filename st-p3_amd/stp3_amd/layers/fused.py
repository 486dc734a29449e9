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


# Test instrument (tests/test_train_parity_gpu.py, never set by the product): float32 tensors on the float32 kernels with
# every operator RESULT (and every gradient that flows back through it) rounded to bf16 -- an emulation of where the bf16
# path rounds, through an independent set of kernels.  A bf16 run that agrees with this emulation differs from the float32
# run by rounding, not by an indexing / padding-lane / accumulation defect (those would be O(1) and in ONE of the two).
EMULATE_BF16 = False


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


# Test instrumentation (tests/test_train_parity_gpu.py, link 2d): the per-sample bias that a pooled-descriptor branch (ASPP
# image pooling, pyramid pooling) contributes to the fused BatchNorm behind it can be recorded in one run and replayed --
# as a constant -- in others: the spatial path of such a block is then compared between two evaluations WITHOUT the
# ill-conditioned BatchNorm over a dozen pooled vectors in the loop.  None in normal operation.
POOLED_BIAS_TAP = None          # None | ('record', list) | ('replay', iterator)


def pooled_bias(t):
    tap = POOLED_BIAS_TAP
    if tap is None or t is None:
        return t
    mode, store = tap
    if mode == 'record':
        store.append(t.detach().clone())
        return t
    return next(store).to(device=t.device, dtype=t.dtype)


def _emu(x):
    return _RoundBf16.apply(x) if (EMULATE_BF16 and torch.is_tensor(x) and x.dtype == torch.float32) else x


def _sync_world(bn):
    if not (bn.training and dist.is_available() and dist.is_initialized()):
        return 1
    if getattr(bn, 'stp3_local_stats', False):
        return 1
    world = dist.get_world_size()
    return 2 if (world == 1 and ops.FORCE_EXCHANGE) else world      # (callers ask "> 1": do the layers exchange?)


def _act(act, y):
    if act == ACT_RELU:
        return F.relu(y)
    if act == ACT_SWISH:
        return F.silu(y)
    return y


def _bn_act_reference_steps(bn, x, act, res, res_mode, sbias, oscale):
    """The plain-torch statement as a generator: yields the packed local [sum | sum of squares] when the statistics are
    shared between replicas and is sent back their (differentiable) sum over the replicas."""
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
            packed = yield torch.cat([s, q])
            s, q = packed[:s.numel()], packed[s.numel():]
            count *= dist.get_world_size()
        mean = s / count
        var = (q / count - mean * mean).clamp_min(0.0)
        if bn.training and bn.track_running_stats:
            with torch.no_grad():
                mom = ops.bn_momentum(bn)
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (count / max(count - 1.0, 1.0)), alpha=mom)
                if bn.num_batches_tracked is not None and not ops.RECOMPUTING[0]:
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


def _reference_member(bn, x, act, res, res_mode, sbias, oscale):
    """(generator, finish) of one statement-route member; zero-padded channel lanes are cut off / put back around it."""
    c = bn.num_features
    lanes = x.shape[1]
    if lanes != c:                                  # zero-padded channel lanes (see ``bn_act``)
        x, res = x[:, :c], None if res is None else res[:, :c]
    gen = _bn_act_reference_steps(bn, x, act, res, res_mode, sbias, oscale)
    return gen, (lambda y: y if lanes == c else F.pad(y, [0, 0] * (y.dim() - 2) + [0, lanes - c]))


def _drive_reference(members):
    """Run statement-route members with ONE (autograd-aware) all-reduce for all that ask for an exchange."""
    import torch.distributed.nn.functional as dfn
    waiting, outs = [], [None] * len(members)
    for i, (gen, finish) in enumerate(members):
        try:
            waiting.append((i, gen, finish, next(gen)))
        except StopIteration as done:
            outs[i] = finish(done.value)
    if waiting:
        packed = dfn.all_reduce(torch.cat([w[3] for w in waiting]) if len(waiting) > 1 else waiting[0][3])
        for (i, gen, finish, buf), part in zip(waiting, packed.split([w[3].numel() for w in waiting])):
            try:
                gen.send(part)
                raise RuntimeError('a second exchange in one BatchNorm statement')
            except StopIteration as done:
                outs[i] = finish(done.value)
    return outs


def bn_act_reference(bn, x, act=ACT_NONE, res=None, res_mode=RES_NONE, sbias=None, oscale=None):
    """Plain-torch statement of ``bn_act`` (any device, any rank count; differentiable)."""
    return _drive_reference([_reference_member(bn, x, act, res, res_mode, sbias, oscale)])[0]


flush_batch_counters = ops.flush_batch_counters          # re-export (parallel.FlatAdam, tests)


def bn_act(bn, x, act=ACT_NONE, res=None, res_mode=RES_NONE, sbias=None, oscale=None, out_slot=None):
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
    args = _kernel_args(bn, x, act, res, res_mode, sbias, oscale, out_slot)
    if args is None:
        # CPU tensors, and float64 on any device: the library has float32 / bf16 kernels only; a float64 evaluation (the
        # noise-free truth of the parity tests) takes the torch statement
        assert out_slot is None, 'an output slot needs the kernel route (see slot_ok)'
        return bn_act_reference(bn, x, act, res, res_mode, sbias, oscale)
    x, weight, bias, res, sbias, oscale, rmean, rvar, training, momentum, eps, act, res_mode, group, channels, out_slot = args
    return _emu(ops.bn_act(x, weight, bias, rmean, rvar, training, momentum, eps, act=act, res=res, res_mode=res_mode, sbias=sbias,
                           oscale=oscale, group=group, channels=channels, out_slot=out_slot))


def slot_ok(x):
    """May a ``bn_act`` of ``x`` write into an output slot?  Only where the kernel route runs and returns its own tensor: a bf16
    4-D GPU tensor (float32 under EMULATE_BF16 comes back rounded -- a new tensor -- and CPU / float64 take the statement)."""
    return x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16


def _kernel_args(bn, x, act, res, res_mode, sbias, oscale, out_slot=None):
    """Positional arguments of ``ops._BnAct`` for this layer, or None when the tensor takes the statement route."""
    if not (x.is_cuda and x.dim() == 4) or x.dtype == torch.float64:
        return None
    if torch.is_autocast_enabled() and x.dtype == torch.float32 and x.shape[2] * x.shape[3] > 1:
        # (1 x 1 maps -- the pooled descriptors of the ASPP / pyramid-pooling branches -- stay float32: the cast and its
        # backward would be two launches for a few hundred numbers, and their consumers take float32)
        x = x.to(torch.get_autocast_dtype('cuda'))
    training = bn.training or not bn.track_running_stats
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        ops.bump_batch_counter(bn)
    group = None if _sync_world(bn) > 1 else False
    if res is None:
        res_mode = RES_NONE
    return (x, bn.weight, bn.bias, res, sbias, oscale, bn.running_mean if bn.track_running_stats else None,
            bn.running_var if bn.track_running_stats else None, bool(training),
            ops.bn_momentum(bn), float(bn.eps), int(act), int(res_mode), group,
            bn.num_features if x.shape[1] != bn.num_features else None, out_slot)


def bn_act_group(items):
    """``bn_act`` for SIBLING layers -- parallel branches whose inputs do not depend on each other's outputs: with
    cross-replica statistics (N > 1 ranks) their exchanges travel together, one all-reduce forward and one backward for
    the whole group (``ops_fused._ExchangeGroup``; the statement route batches the same way).  ``items``: dicts with the
    keyword arguments of ``bn_act`` (bn, x, act, res, res_mode, sbias, oscale); a member may also be a
    ('conv_bn_act', args) pair prepared by ``conv_bn_act_member``.  Single-process runs take the ordinary operators."""
    def norm(it):
        return (it['bn'], it['x'], it.get('act', ACT_NONE), it.get('res'), it.get('res_mode', RES_NONE) if it.get('res') is not None
                else RES_NONE, it.get('sbias'), it.get('oscale'), it.get('out_slot'))
    first = items[0]
    bn0 = first['bn'] if isinstance(first, dict) else first[2]
    if _sync_world(bn0) <= 1:
        return [_run_member(it) for it in items]
    if all(isinstance(it, dict) and _takes_statement(it['x']) for it in items):
        return _drive_reference([_reference_member(*norm(it)[:7]) for it in items])
    from .. import ops_fused
    members = []
    for it in items:
        if isinstance(it, dict):
            args = _kernel_args(*norm(it))
            if args is None:
                raise ops._lib.Stp3HipError('bn_act_group: members on the kernels and on the statement route cannot share an exchange')
            members.append(('bn_act', args))
        else:
            members.append((it[0], it[1]))
    return ops_fused.exchange_group(members)


def _takes_statement(x):
    return not (x.is_cuda and x.dim() == 4) or x.dtype == torch.float64


def _run_member(it):
    if isinstance(it, dict):
        return bn_act(it['bn'], it['x'], it.get('act', ACT_NONE), it.get('res'), it.get('res_mode', RES_NONE), it.get('sbias'),
                      it.get('oscale'), it.get('out_slot'))
    from .. import ops_fused
    return ops_fused._ConvBnAct.apply(*it[1])


def conv_bn_act_member(x, conv, bn, act):
    """A conv -> BatchNorm -> activation layer as a member of ``bn_act_group``: the fused operator when the layer
    qualifies (``_fusable_conv_bn``), otherwise the convolution now and its BatchNorm as the member."""
    if _fusable_conv_bn(conv, bn, x):
        from .. import ops_fused
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            ops.bump_batch_counter(bn)
        group = None if _sync_world(bn) > 1 else False
        return ('conv_bn_act', (x, conv.weight, conv.bias, bn.weight, bn.bias, None, bn.running_mean, bn.running_var,
                                ops.bn_momentum(bn), float(bn.eps), int(act),
                                int(RES_NONE), ops._pair(conv.stride)[0], ops._pair(conv.padding), ops._pair(conv.dilation),
                                group, None), bn)
    return dict(bn=bn, x=conv_module(conv, x), act=act)


# Dense convolutions.  Every convolution on the GPU -- forward, data gradient and weight gradient, all layer shapes of the
# model, the 3-channel stem included (its channels are zero-padded to 8) -- runs on the hand-written MFMA implicit-GEMM
# kernels of stp3_conv.hip: bf16 tensors and anything under autocast directly (the benchmarked path), FLOAT32 tensors
# outside autocast (the float32 legs of the parity tests) through the three-term bf16 split of ``ops.conv2d_f32`` --
# float32-accurate on the same kernels, so that the tight float32 / float64 parity chain pins ``conv2d_igemm_kernel`` and
# ``conv2d_wgrad_kernel`` at the step's real shapes.  No vendor convolution on the GPU; CPU tensors and float64 take torch's.
def _use_mfma(x, weight, stride):
    if not x.is_cuda:
        return False
    if not (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()):
        return False
    return ops.conv2d_supported(x, weight, stride)


def _use_mfma_f32(x, weight, stride):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and not torch.is_autocast_enabled()
            and ops.conv2d_supported(x, weight, stride))


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    if _use_mfma(x, weight, stride):
        return ops.conv2d(x, weight, bias, stride, padding, dilation)
    if _use_mfma_f32(x, weight, stride):
        if EMULATE_BF16:                      # bf16 operands (weights rounded like the shadows), float32 accumulation, bf16 result
            return _emu(ops.conv2d_f32(x, _emu(weight), bias, stride, padding, dilation))
        return ops.conv2d_f32(x, weight, bias, stride, padding, dilation)
    return F.conv2d(x, weight, bias, stride, padding, dilation)


class _PlaneMean(torch.autograd.Function):
    """Mean over the plane of every (sample, channel) of an (N, C, H, W) tensor, float32 result (N, C).  torch's own
    ``mean`` / ``AdaptiveAvgPool2d(1)`` give the same value, but their backward hands back the broadcast gradient in
    NCHW-contiguous memory whatever the layout of x; for a channels-last x that puts every later addition of input
    gradients on torch's strided element-wise kernel (3x slower).  Here the gradient is laid out like x."""

    @staticmethod
    def forward(ctx, x):
        cl = x.is_contiguous(memory_format=torch.channels_last)
        ctx.meta = (tuple(x.shape), x.dtype, cl)
        if (x.is_cuda and cl and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32)
                and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.shape[2] * x.shape[3] > 1):
            # per-(sample, channel) sums through the squeeze kernel of the MBConv blocks (stp3_se_pool: coalesced channel vectors,
            # float32 accumulation, deterministic two-stage sum): 38 -> ~17 us for a (12, 64, 200, 200) tensor, where torch's
            # reduction of a channels-last tensor over (H, W) runs at 1.6 TB/s
            from .. import ops_fused
            xv, dims = ops_fused._se_dims(x)
            return ops_fused._se_pool(xv, dims) * (1.0 / (x.shape[2] * x.shape[3]))
        return x.mean(dim=(2, 3), dtype=torch.float64 if x.dtype == torch.float64 else torch.float32)

    @staticmethod
    def backward(ctx, g):
        (n, c, h, w), dtype, cl = ctx.meta
        if g.is_cuda and g.dtype == torch.float32 and dtype == torch.bfloat16:
            # (scale and round in ONE launch: float32 product, one rounding -- what the two operators below give)
            g = torch.mul(g, 1.0 / (h * w), out=torch.empty(g.shape, dtype=dtype, device=g.device))
        else:
            g = (g * (1.0 / (h * w))).to(dtype)
        g = g.view(n, c, 1, 1).expand(n, c, h, w)
        if g.is_cuda and cl and dtype in (torch.bfloat16, torch.float32):
            # the broadcast stays a stride-0 VIEW: the single-pass sum of the input's gradients (ops.fan_out) adds it as a
            # per-(sample, channel) term; any other consumer sees an ordinary (expanded) tensor
            return g
        return g.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)


def plane_mean(x):
    """(N, C, H, W) -> (N, C) float32 mean over H x W (see ``_PlaneMean``)."""
    return _PlaneMean.apply(x)


def conv1x1_on_vector(x, weight, bias=None):
    """A 1x1 convolution of a (N, C, 1, 1) map (or (N, C, T, 1, 1) with a 1x1x1 kernel) is a plain GEMM.

    Under GPU autocast it is evaluated in FLOAT32 with autocast off: these are (12, 128) x (128, 128)-sized products whose
    time is launch latency, and autocast wrapped each of them in three cast kernels forward (operand, weight, result back to
    float32 for the per-sample bias it becomes) and as many backward -- ~100 launches per step for ten products.  The pooled
    descriptors arrive in float32 (``plane_mean``) and the weights are float32 parameters: nothing is cast at all."""
    if x.is_cuda and torch.is_autocast_enabled():
        with torch.autocast('cuda', enabled=False):
            return _conv1x1_on_vector(x.float(), weight.float(), None if bias is None else bias.float())
    if x.dtype != weight.dtype:
        x = x.to(weight.dtype)
    return _conv1x1_on_vector(x, weight, bias)


def _linear(x2, w2, bias):
    """F.linear on the rows of x2 -- through stp3_linear_* (one launch forward, one backward) where it applies: the vendor GEMM
    takes one launch forward and two backward, 9-19 us each, 38 per training step for these products."""
    if ops.small_linear_supported(x2, w2, bias):
        return ops.small_linear(x2, w2, bias)
    return F.linear(x2, w2, bias)


def _conv1x1_on_vector(x, weight, bias):
    w2 = weight.flatten(1)
    if x.dim() == 5:
        n, c, t = x.shape[:3]
        y = _linear(x.reshape(n, c, t).transpose(1, 2).reshape(n * t, c), w2, bias).view(n, t, -1)     # (N, T, Co)
        return y.transpose(1, 2).reshape(n, -1, t, 1, 1)
    y = _linear(x.flatten(1), w2, bias)
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


# conv -> BatchNorm (-> activation, + skip) layers written out in a module's forward (the ResNet blocks of the decoder, its stem
# and its upsample-add stages) through the operator whose convolution epilogue produces the statistics, like ``run_fused`` does for
# Sequentials.  Off: the convolution, then the BatchNorm with its own statistics pass over the convolution output.
FUSE_WRITTEN_OUT_LAYERS = True


def conv_bn_act_layer(x, conv, bn, act, res=None, res_mode=RES_NONE):
    """act(bn(conv(x))) (+ res: ``res_mode``) -- ``ops_fused.conv_bn_act`` where the layer qualifies (``_fusable_conv_bn``)."""
    if FUSE_WRITTEN_OUT_LAYERS and _fusable_conv_bn(conv, bn, x):
        from .. import ops_fused
        group = None if _sync_world(bn) > 1 else False
        return ops_fused.conv_bn_act(x, conv.weight, conv.bias, bn, act, res, res_mode if res is not None else RES_NONE,
                                     conv.stride, conv.padding, conv.dilation, group=group)
    return bn_act(bn, conv_module(conv, x), act, res=res, res_mode=res_mode)


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
