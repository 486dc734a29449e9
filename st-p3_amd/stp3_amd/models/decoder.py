"""BEV decoder: 7x7/2 stem + ResNet-18 stages 1-3, three upsample-add skips, per-task heads.
Mirrors ``stp3/models/decoder.py`` (Decoder :8-140): same constructor, parameter names and output
dictionary (``None`` for disabled heads)."""
import torch
import torch.nn as nn

from .. import ops
from ..layers.convolutions import UpsamplingAdd
from ..layers.fused import ACT_RELU, _sync_world, bn_act, bn_act_group, conv_bn_act_layer, conv_bn_act_member, conv_module, run_fused
from .resnet import resnet18


def _head(channels, out_channels, sigmoid=False):
    mods = [nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(channels),
            nn.ReLU(inplace=True), nn.Conv2d(channels, out_channels, kernel_size=1, padding=0)]
    if sigmoid:
        mods.append(nn.Sigmoid())
    return nn.Sequential(*mods)


class _SelectFrame(torch.autograd.Function):
    """``x.view(b, s, C, H, W)[:, idx]`` of a frame-folded (b*s, C, H, W) tensor.  Plain indexing gives the same values;
    its backward, however, builds the zero-padded gradient as a contiguous 5-D tensor, i.e. in a different memory layout
    than the channels-last gradients of the other five heads it is added to (each such addition then runs torch's
    generic strided kernel: 93 instead of 27 us at 12 x 64 x 200 x 200).  Here the padded gradient has x's layout."""

    @staticmethod
    def forward(ctx, x, b, s, idx):
        ctx.cfg = (b, s, idx)
        return x.view(b, s, *x.shape[1:])[:, idx].clone()       # (a custom Function must not hand back a view of its input)

    @staticmethod
    def backward(ctx, g):
        b, s, idx = ctx.cfg
        out = torch.empty((b * s,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device,
                          memory_format=torch.channels_last if g.is_cuda else torch.contiguous_format).zero_()
        out.view(b, s, *g.shape[1:])[:, idx] = g
        return out, None, None, None


# The heads that read the SAME tensor run their first layers (3x3 convolution -> BatchNorm -> ReLU, decoder.py:42-66) as
# ONE convolution with the output channels of all heads side by side and ONE BatchNorm over those channels, and their
# 1x1 output convolutions as ONE convolution with a block-diagonal weight: exact (a BatchNorm is per channel, a
# convolution per output channel), the operand is staged once instead of once per head, five data gradients and their
# additions become one.  Training mode on the bf16 kernels; everything else takes the heads one by one.
MERGE_HEADS = True


class Decoder(nn.Module):
    def __init__(self, in_channels, n_classes, n_present, n_hdmap, predict_gate):
        super().__init__()
        self.perceive_hdmap = predict_gate['perceive_hdmap']
        self.predict_pedestrian = predict_gate['predict_pedestrian']
        self.predict_instance = predict_gate['predict_instance']
        self.predict_future_flow = predict_gate['predict_future_flow']
        self.planning = predict_gate['planning']
        self.n_classes = n_classes
        self.n_present = n_present
        if self.predict_instance is False and self.predict_future_flow is True:
            raise ValueError('flow cannot be True when not predicting instance')

        backbone = resnet18(pretrained=False, zero_init_residual=True)
        self.first_conv = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1, self.relu = backbone.bn1, backbone.relu
        self.layer1, self.layer2, self.layer3 = backbone.layer1, backbone.layer2, backbone.layer3

        shared = in_channels
        self.up3_skip = UpsamplingAdd(256, 128, scale_factor=2)
        self.up2_skip = UpsamplingAdd(128, 64, scale_factor=2)
        self.up1_skip = UpsamplingAdd(64, shared, scale_factor=2)

        self.segmentation_head = _head(shared, n_classes)
        if self.predict_pedestrian:
            self.pedestrian_head = _head(shared, n_classes)
        if self.perceive_hdmap:
            self.hdmap_head = _head(shared, 2 * n_hdmap)
        if self.predict_instance:
            self.instance_offset_head = _head(shared, 2)
            self.instance_center_head = _head(shared, 1, sigmoid=True)
        if self.predict_future_flow:
            self.instance_future_head = _head(shared, 2)
        if self.planning:
            self.costvolume_head = _head(shared, 1)

    def _merged_heads(self, heads, x, others=()):
        """{name: output} of the heads in ``heads`` (all reading x) through the merged operators and of ``others`` (name,
        head, input: heads on another tensor -- the hd-map head reads the present frame only) one by one, or None when the
        configuration does not qualify (see MERGE_HEADS).  With cross-replica statistics the merged BatchNorm and the
        first BatchNorms of ``others`` are siblings: one exchange per pass for all of them."""
        from ..layers.fused import _fusable_conv_bn, conv2d
        from .. import ops_fused
        if len(heads) < 2:
            return None
        convs, bns, lasts = [h[0] for _, h in heads], [h[1] for _, h in heads], [h[3] for _, h in heads]
        c = convs[0].out_channels
        if not all(_fusable_conv_bn(cv, bn, x) and cv.out_channels == c and cv.kernel_size == (3, 3) and cv.bias is None
                   and bn.momentum == bns[0].momentum and bn.eps == bns[0].eps and bn.affine
                   for cv, bn in zip(convs, bns)) or c % 8:
            return None
        # the merged BatchNorm updates float32 statistics in one packed buffer: a module converted to another dtype
        # (``.half()`` / ``.bfloat16()``) keeps its own buffers and runs the heads one by one
        if any(bn.running_mean is None or bn.running_mean.dtype != torch.float32 or bn.running_var.dtype != torch.float32
               for bn in bns):
            return None
        running_mean, running_var = self._packed_running_stats(bns)
        for bn in bns:
            if bn.num_batches_tracked is not None:
                ops.bump_batch_counter(bn)
        shared = _sync_world(bns[0]) > 1
        assembled = ops.assembled_weight_supported(x, [cv.weight for cv in convs] + [m.weight for m in lasts]) and \
            all(m.bias is not None for m in lasts)
        if assembled:
            # the heads' kernels one below the other: pieces of ONE weight (ops.assembled_weight: no concatenation per step,
            # and the gradient of the whole is cut into the heads' gradients by the launch that serves every such weight)
            w1 = ops.assembled_weight((id(self), 'heads 3x3'), (len(convs) * c, convs[0].weight.shape[1], 3, 3),
                                      [ops.weight_piece(cv.weight, cv.weight.detach(), k * c, 0) for k, cv in enumerate(convs)])
        else:
            w1 = torch.cat([cv.weight for cv in convs], dim=0)                           # (heads * C, Cin, 3, 3)
        gamma, beta = torch.cat([bn.weight for bn in bns]), torch.cat([bn.bias for bn in bns])
        args = (x, w1, None, gamma, beta, None, running_mean, running_var,
                ops.bn_momentum(bns[0]), float(bns[0].eps), int(ACT_RELU),
                int(ops.RES_NONE), 1, (1, 1), (1, 1), None if shared else False, None)
        out = {}
        if shared and others:
            mids = bn_act_group([('conv_bn_act', args, bns[0])]
                                + [conv_bn_act_member(inp, head[0], head[1], ACT_RELU) for _, head, inp in others])
            mid = mids[0]
            for (name, head, _), m in zip(others, mids[1:]):
                out[name] = run_fused(list(head)[3:], m)
        else:
            mid = ops_fused._ConvBnAct.apply(*args)
            for name, head, inp in others:
                out[name] = run_fused(head, inp)
        # second layers: head k maps its C channels of `mid` to its outputs -- a block-diagonal 1x1 convolution
        n_out = sum(m.out_channels for m in lasts)
        if assembled:
            # (output channels padded to 8 lanes: both gradients of the layer then run without padded copies of dy and the weight)
            lanes_out = (n_out + 7) // 8 * 8
            pieces, co = [], 0
            for k, m in enumerate(lasts):
                pieces.append(ops.weight_piece(m.weight, m.weight.detach(), co, k * c))
                co += m.out_channels
            w2 = ops.assembled_weight((id(self), 'heads 1x1'), (lanes_out, len(lasts) * c, 1, 1), pieces)
            b2 = torch.cat([m.bias for m in lasts] + ([m.bias.new_zeros(lanes_out - n_out)] if lanes_out != n_out else []))
            y = conv2d(mid, w2, b2)[:, :n_out]
        else:
            w2 = torch.block_diag(*[m.weight.flatten(1) for m in lasts])[:, :, None, None]    # (sum of outputs, heads * C, 1, 1)
            b2 = torch.cat([m.bias for m in lasts])
            y = conv2d(mid, w2, b2)
        # (one split: its backward writes the heads' gradients into ONE tensor, instead of a zero-fill, a copy and an addition per head)
        for (name, head), t in zip(heads, y.split([m.out_channels for m in lasts], dim=1)):
            for extra in list(head)[4:]:                                 # the sigmoid of the centerness head
                t = extra(t)
            out[name] = t
        return out

    def _packed_running_stats(self, bns):
        """The running statistics of the heads' BatchNorms as slices of ONE buffer each (the merged BatchNorm updates them
        in place, all heads at once): the modules' own buffers are re-pointed into it (same values, same names, same
        state-dict entries); re-packed whenever the modules were moved or reloaded out of it."""
        packed = self.__dict__.get('_packed_stats')
        c = bns[0].num_features
        ok = packed is not None and packed[0].numel() == c * len(bns) and packed[0].device == bns[0].running_mean.device
        if ok:
            for k, bn in enumerate(bns):
                ok = ok and bn.running_mean.data_ptr() == packed[0].data_ptr() + 4 * c * k \
                    and bn.running_var.data_ptr() == packed[1].data_ptr() + 4 * c * k
        if not ok:
            with torch.no_grad():
                packed = (torch.cat([bn.running_mean.float() for bn in bns]), torch.cat([bn.running_var.float() for bn in bns]))
                for k, bn in enumerate(bns):
                    bn.running_mean.data = packed[0][k * c:(k + 1) * c]
                    bn.running_var.data = packed[1][k * c:(k + 1) * c]
            self.__dict__['_packed_stats'] = packed
        return packed

    def forward(self, x):
        b, s, c, h, w = x.shape
        x = x.reshape(b * s, c, h, w)
        skip1 = x
        x = self.layer1(conv_bn_act_layer(x, self.first_conv, self.bn1, ACT_RELU))  # 1/2
        skip2 = x
        x = self.layer2(x)                                            # 1/4
        skip3 = x
        x = self.layer3(x)                                            # 1/8
        x = self.up3_skip(x, skip3)
        x = self.up2_skip(x, skip2)
        x = self.up1_skip(x, skip1)

        def per_frame(t):
            return None if t is None else t.view(b, s, *t.shape[1:])

        # every head reads x: their input gradients are added in one pass (ops.fan_out), not pairwise
        n_heads = (1 + int(self.predict_pedestrian) + int(self.perceive_hdmap) + 2 * int(self.predict_instance)
                   + int(self.predict_future_flow) + int(self.planning))
        xs = iter(ops.fan_out(x, n_heads) if x.is_cuda else [x] * n_heads)
        heads = [('segmentation', self.segmentation_head, next(xs))]
        if self.predict_pedestrian:
            heads.append(('pedestrian', self.pedestrian_head, next(xs)))
        if self.perceive_hdmap:
            heads.append(('hdmap', self.hdmap_head, _SelectFrame.apply(next(xs), b, s, self.n_present - 1)))   # decoder.py:122
        if self.predict_instance:
            heads += [('instance_center', self.instance_center_head, next(xs)),
                      ('instance_offset', self.instance_offset_head, next(xs))]
        if self.predict_future_flow:
            heads.append(('instance_flow', self.instance_future_head, next(xs)))
        if self.planning:
            heads.append(('costvolume', self.costvolume_head, next(xs)))
        others = [(name, head, inp) for name, head, inp in heads if name == 'hdmap']
        # (the merged operator reads x through ONE of the fan-out handles, so that its input gradient and the hd-map head's meet
        # in the single-pass sum of ops.fan_out instead of a strided copy + an addition by the autograd engine)
        x_merged = next((inp for name, _, inp in heads if name != 'hdmap'), x)
        merged = self._merged_heads([(name, head) for name, head, inp in heads if name != 'hdmap'], x_merged, others) if MERGE_HEADS else None
        if merged is not None:
            # (the handles of ops.fan_out that the merged heads did not take are simply dropped: no gradient arrives there)
            out = merged
        elif _sync_world(self.segmentation_head[1]) > 1:
            # N > 1 ranks: the heads are siblings -- the statistics of their first BatchNorms travel in ONE exchange per
            # pass; the 1x1 output convolutions (and the sigmoid of the centerness head) follow per head
            mids = bn_act_group([conv_bn_act_member(inp, head[0], head[1], ACT_RELU) for _, head, inp in heads])
            out = {name: run_fused(list(head)[3:], mid) for (name, head, _), mid in zip(heads, mids)}
        else:
            out = {name: run_fused(head, inp) for name, head, inp in heads}
        return {
            'segmentation': per_frame(out['segmentation']),
            'pedestrian': per_frame(out.get('pedestrian')),
            'hdmap': out.get('hdmap'),
            'instance_center': per_frame(out.get('instance_center')),
            'instance_offset': per_frame(out.get('instance_offset')),
            'instance_flow': per_frame(out.get('instance_flow')),
            'costvolume': per_frame(out['costvolume'].squeeze(1) if self.planning else None),
        }
