"""ResNet-18 stages 1-3 as the reference's BEV ``Decoder`` uses them.

The reference takes ``bn1, relu, layer1, layer2, layer3`` from
``torchvision.models.resnet.resnet18(pretrained=False, zero_init_residual=True)``
(stp3/models/decoder.py:3, 22-30).  torchvision 0.11.3 is not vendored or installed, so the
published architecture (He et al. 2016) is restated here with torchvision's parameter names
(``layerK.J.conv1/bn1/conv2/bn2/downsample.0/downsample.1``) and its initialisation (Kaiming
normal fan_out, last BN gamma of every block zero).  PARITY UNPINNED against the real package.
"""
import torch.nn as nn

from ..layers import fused
from ..layers.fused import ACT_NONE, ACT_RELU, RES_BEFORE_ACT, bn_act_group, conv_bn_act_layer, conv_bn_act_member, conv_module


class BasicBlock(nn.Module):
    def __init__(self, in_ch, out_ch, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_ch)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_ch)
        self.downsample = None
        if stride != 1 or in_ch != out_ch:
            self.downsample = nn.Sequential(nn.Conv2d(in_ch, out_ch, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(out_ch))

    def forward(self, x):
        # every conv -> BatchNorm pair through the operator whose convolution epilogue produces the statistics
        # (fused.conv_bn_act_layer / conv_bn_act_member): no statistics pass over the convolution outputs
        if self.downsample is None:
            skip = x
            y = conv_bn_act_layer(x, self.conv1, self.bn1, ACT_RELU)
        elif fused.FUSE_WRITTEN_OUT_LAYERS:
            # the down-sampling skip and the first convolution read the same tensor: sibling BatchNorms, one statistics
            # exchange for both when the statistics are shared between ranks (fused.bn_act_group)
            skip, y = bn_act_group([conv_bn_act_member(x, self.downsample[0], self.downsample[1], ACT_NONE),
                                    conv_bn_act_member(x, self.conv1, self.bn1, ACT_RELU)])
        else:
            skip, y = bn_act_group([dict(bn=self.downsample[1], x=conv_module(self.downsample[0], x), act=ACT_NONE),
                                    dict(bn=self.bn1, x=conv_module(self.conv1, x), act=ACT_RELU)])
        return conv_bn_act_layer(y, self.conv2, self.bn2, ACT_RELU, res=skip, res_mode=RES_BEFORE_ACT)


class ResNet18Stages(nn.Module):
    """Exposes ``bn1, relu, layer1, layer2, layer3`` (all the decoder uses)."""

    def __init__(self, zero_init_residual=True):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0.0)


def resnet18(pretrained=False, zero_init_residual=True):
    assert not pretrained, 'no network: pretrained ImageNet weights are unavailable'
    return ResNet18Stages(zero_init_residual=zero_init_residual)
