"""BEV decoder: 7x7/2 stem + ResNet-18 stages 1-3, three upsample-add skips, per-task heads.
Mirrors ``stp3/models/decoder.py`` (Decoder :8-140): same constructor, parameter names and output
dictionary (``None`` for disabled heads)."""
import torch
import torch.nn as nn

from ..layers.convolutions import UpsamplingAdd
from ..layers.fused import ACT_RELU, _sync_world, bn_act, bn_act_group, conv_bn_act_member, conv_module, run_fused
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

    def forward(self, x):
        b, s, c, h, w = x.shape
        x = x.reshape(b * s, c, h, w)
        skip1 = x
        x = self.layer1(bn_act(self.bn1, conv_module(self.first_conv, x), ACT_RELU))  # 1/2
        skip2 = x
        x = self.layer2(x)                                            # 1/4
        skip3 = x
        x = self.layer3(x)                                            # 1/8
        x = self.up3_skip(x, skip3)
        x = self.up2_skip(x, skip2)
        x = self.up1_skip(x, skip1)

        def per_frame(t):
            return None if t is None else t.view(b, s, *t.shape[1:])

        present = _SelectFrame.apply(x, b, s, self.n_present - 1)      # decoder.py:122
        heads = [('segmentation', self.segmentation_head, x)]
        if self.predict_pedestrian:
            heads.append(('pedestrian', self.pedestrian_head, x))
        if self.perceive_hdmap:
            heads.append(('hdmap', self.hdmap_head, present))
        if self.predict_instance:
            heads += [('instance_center', self.instance_center_head, x), ('instance_offset', self.instance_offset_head, x)]
        if self.predict_future_flow:
            heads.append(('instance_flow', self.instance_future_head, x))
        if self.planning:
            heads.append(('costvolume', self.costvolume_head, x))
        if _sync_world(self.segmentation_head[1]) > 1:
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
