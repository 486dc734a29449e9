"""Per-camera image encoder: EfficientNet trunk -> two DeepLabHead + UpsamplingConcat heads giving
C feature channels and D depth logits at 1/8 resolution.  Mirrors the reference's
``stp3/models/encoder.py`` (Encoder :9-35, delete_unused_layers :39-55, get_features_depth
:57-97) -- same constructor, attributes and parameter names."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..layers.convolutions import DeepLabHead, UpsamplingConcat
from ..layers.fused import ACT_SWISH, bn_act
from .efficientnet import EfficientNet

_REDUCTION_CHANNELS = {'b4': [0, 24, 32, 56, 160, 448], 'b0': [0, 16, 24, 40, 112, 320]}
_LAST_BLOCK = {'b4': 21, 'b0': 10}          # last trunk block kept when downsample == 8 (encoder.py:43-46)


def _recomputed(block, x, rate, drop_scale):
    """``block(x)`` keeping only x for the backward pass: the block's forward is run again when its gradients are needed
    (torch.utils.checkpoint, non-reentrant: the saved tensors of the block's operators are rebuilt on demand).  The second
    run happens under ``ops.recomputing()``: BatchNorm running statistics and batch counters are not touched again; the
    drop-connect factors are an argument, so both runs see the same ones; the kernels are deterministic, so the rebuilt
    activations are the ones the first run made."""
    import contextlib
    from torch.utils.checkpoint import checkpoint
    return checkpoint(lambda t: block(t, drop_connect_rate=rate, drop_scale=drop_scale), x, use_reentrant=False,
                      context_fn=lambda: (contextlib.nullcontext(), ops.recomputing()))


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
        # recompute the MBConv blocks' activations in the backward pass instead of keeping them (``_recomputed``): off by
        # default; what makes BASELINE configs[4]'s four samples per GPU of 896 x 1600 images fit (scripts/run_c5_step.py)
        self.recompute_blocks = False
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

    def _drop_connect_scales(self, x, rates):
        """Drop-connect factors floor(keep + u) / keep of every residual block, u ~ U[0,1)^N for all blocks drawn with ONE
        generator call and the arithmetic on all of them at once: 3 small launches per step instead of 3 per block (17
        generator launches when drawn block by block; the device generator's stream is no part of any contract -- the
        reference draws on the GPU, efficientnet_pytorch utils.drop_connect)."""
        which = [i for i, b in enumerate(self.backbone._blocks)
                 if rates[i] and b.stride == 1 and b.in_ch == b.out_ch]
        if not which:
            return {}
        key = (x.device, tuple(which), tuple(rates))
        if getattr(self, '_keep_cache', (None, None))[0] != key:
            self._keep_cache = (key, torch.tensor([1.0 - rates[i] for i in which], dtype=torch.float32,
                                                  device=x.device)[:, None])
        keep = self._keep_cache[1]
        u = torch.rand(len(which), x.shape[0], dtype=torch.float32, device=x.device)
        scale = torch.floor(keep + u) / keep
        return {i: scale[k] for k, i in enumerate(which)}

    def trunk(self, x):
        """Stem + MBConv blocks; returns the reduction endpoints (tensor before every resolution drop,
        plus the final tensor), encoder.py:59-82."""
        bb = self.backbone
        endpoints = []
        x = bn_act(bb._bn0, bb._conv_stem(x), ACT_SWISH)
        n_blocks = len(bb._blocks)
        base_rate = bb._global_params.drop_connect_rate
        rates = [base_rate * float(idx) / n_blocks if base_rate else base_rate for idx in range(n_blocks)]
        scales = self._drop_connect_scales(x, rates) if (self.training and x.is_cuda) else {}
        recompute = self.recompute_blocks and self.training and torch.is_grad_enabled()
        for idx, block in enumerate(bb._blocks):
            if recompute and x.requires_grad:
                y = _recomputed(block, x, rates[idx], scales.get(idx))
            else:
                y = block(x, drop_connect_rate=rates[idx], drop_scale=scales.get(idx))
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
