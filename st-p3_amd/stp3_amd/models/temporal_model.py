"""Temporal fusion over the T BEV frames: (T-1) TemporalBlocks then a DeepLabHead applied per
frame.  Mirrors ``stp3/models/temporal_model.py`` (TemporalModel :7-60, TemporalModelIdentity
:63-70)."""
import torch.nn as nn

from ..layers.convolutions import DeepLabHead
from ..layers.temporal import Bottleneck3D, TemporalBlock


class TemporalModel(nn.Module):
    def __init__(self, in_channels, receptive_field, input_shape, start_out_channels=64, extra_in_channels=0,
                 n_spatial_layers_between_temporal_layers=0, use_pyramid_pooling=True):
        super().__init__()
        self.receptive_field = receptive_field
        h, w = input_shape
        blocks = []
        block_in, block_out = in_channels, start_out_channels
        for _ in range(receptive_field - 1):
            blocks.append(TemporalBlock(block_in, block_out, use_pyramid_pooling=bool(use_pyramid_pooling),
                                        pool_sizes=[(2, h, w)] if use_pyramid_pooling else None))
            # INBETWEEN_LAYERS spatial bottlenecks behind every temporal block (temporal_model.py:33-37; 0 in every shipped config)
            blocks.extend(Bottleneck3D(block_out, block_out, kernel_size=(1, 3, 3))
                          for _ in range(n_spatial_layers_between_temporal_layers))
            block_in = block_out
            block_out += extra_in_channels
        self.out_channels = block_in
        self.final_conv = DeepLabHead(block_out, block_out, hidden_channel=128)
        self.model = nn.Sequential(*blocks)

    def forward(self, x, extra=None):
        """(B, T, C, X, Y) -> (B, T, C', X, Y).  ``extra`` (B, T, E): the last E input channels as per-frame
        constants (the six ego-motion planes of stp3.py:145-152) -- folded into the first block, never materialised."""
        x = x.permute(0, 2, 1, 3, 4)                        # (B, C, T, X, Y)
        for i, blk in enumerate(self.model):
            if x.is_cuda and isinstance(blk, Bottleneck3D):
                x = x.contiguous()                           # (plain torch modules: their own layout)
            x = blk(x, extra.permute(0, 2, 1)) if (i == 0 and extra is not None) else blk(x)
        x = x.permute(0, 2, 1, 3, 4)
        b, s, c, h, w = x.shape
        x = self.final_conv(x.reshape(b * s, c, h, w))
        return x.view(b, s, c, h, w)


class TemporalModelIdentity(nn.Module):
    def __init__(self, in_channels, receptive_field):
        super().__init__()
        self.receptive_field = receptive_field
        self.out_channels = in_channels

    def forward(self, x):
        return x
