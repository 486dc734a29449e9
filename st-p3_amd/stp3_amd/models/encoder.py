"""Per-camera image encoder: EfficientNet trunk -> two DeepLabHead + UpsamplingConcat heads giving
C feature channels and D depth logits at 1/8 resolution.  Mirrors the reference's
``stp3/models/encoder.py`` (Encoder :9-35, delete_unused_layers :39-55, get_features_depth
:57-97) -- same constructor, attributes and parameter names."""
import math

import torch.nn as nn

from ..layers.convolutions import DeepLabHead, UpsamplingConcat
from ..layers.fused import ACT_SWISH, bn_act
from .efficientnet import EfficientNet

_REDUCTION_CHANNELS = {'b4': [0, 24, 32, 56, 160, 448], 'b0': [0, 16, 24, 40, 112, 320]}
_LAST_BLOCK = {'b4': 21, 'b0': 10}          # last trunk block kept when downsample == 8 (encoder.py:43-46)


class Encoder(nn.Module):
    def __init__(self, cfg, D):
        super().__init__()
        self.D = D
        self.C = cfg.OUT_CHANNELS
        self.use_depth_distribution = cfg.USE_DEPTH_DISTRIBUTION
        self.downsample = cfg.DOWNSAMPLE
        self.version = cfg.NAME.split('-')[1]
        if self.version not in _REDUCTION_CHANNELS:
            raise NotImplementedError(cfg.NAME)
        self.backbone = EfficientNet.from_pretrained(cfg.NAME)
        self.delete_unused_layers()
        self.reduction_channel = _REDUCTION_CHANNELS[self.version]
        self.upsampling_out_channel = [0, 48, 64, 128, 512]
        index = int(math.log2(self.downsample))
        self._index = index
        deep, skip = self.reduction_channel[index + 1], self.reduction_channel[index]
        if self.use_depth_distribution:
            self.depth_layer_1 = DeepLabHead(deep, deep, hidden_channel=64)
            self.depth_layer_2 = UpsamplingConcat(deep + skip, self.D)
        self.feature_layer_1 = DeepLabHead(deep, deep, hidden_channel=64)
        self.feature_layer_2 = UpsamplingConcat(deep + skip, self.C)

    def delete_unused_layers(self):
        if self.downsample == 8:
            keep = _LAST_BLOCK[self.version] + 1
            for idx in reversed(range(keep, len(self.backbone._blocks))):
                del self.backbone._blocks[idx]
        for name in ('_conv_head', '_bn1', '_avg_pooling', '_dropout', '_fc'):
            delattr(self.backbone, name)

    def trunk(self, x):
        """Stem + MBConv blocks; returns the reduction endpoints (tensor before every resolution drop,
        plus the final tensor), encoder.py:59-82."""
        bb = self.backbone
        endpoints = []
        x = bn_act(bb._bn0, bb._conv_stem(x), ACT_SWISH)
        n_blocks = len(bb._blocks)
        base_rate = bb._global_params.drop_connect_rate
        for idx, block in enumerate(bb._blocks):
            rate = base_rate * float(idx) / n_blocks if base_rate else base_rate
            y = block(x, drop_connect_rate=rate)
            if x.size(2) > y.size(2):
                endpoints.append(x)
            x = y
            if self.downsample == 8 and idx == _LAST_BLOCK[self.version]:
                break
        endpoints.append(x)
        return endpoints

    def get_features_depth(self, x):
        endpoints = self.trunk(x)
        deep, skip = endpoints[self._index], endpoints[self._index - 1]   # reduction_{index+1}, reduction_{index}
        feature = self.feature_layer_2(self.feature_layer_1(deep), skip)
        depth = None
        if self.use_depth_distribution:
            depth = self.depth_layer_2(self.depth_layer_1(deep), skip)
        return feature, depth

    def forward(self, x):
        return self.get_features_depth(x)
