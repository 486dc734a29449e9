"""Future prediction: ``stp3/models/future_prediction.py:7-46`` -- a Dual_GRU rolls the present state forward
``n_future`` steps from the latent sample, then ``n_gru_blocks`` SpatialGRU + residual stages run over the whole
(past + future) sequence."""
import torch
import torch.nn as nn

from ..layers.convolutions import Block, DeepLabHead
from ..layers.temporal import Dual_GRU, SpatialGRU, batch_major, frames_as_batch, stack_frames, unbind_frames


class FuturePrediction(nn.Module):
    def __init__(self, in_channels, latent_dim, n_future, mixture=True, n_gru_blocks=2, n_res_layers=1):
        super().__init__()
        self.n_spatial_gru = n_gru_blocks
        self.dual_grus = Dual_GRU(latent_dim, in_channels, n_future=n_future, mixture=mixture)
        self.res_blocks1 = nn.Sequential(*[Block(in_channels) for _ in range(n_res_layers)])
        grus, res = [], []
        for i in range(n_gru_blocks):
            grus.append(SpatialGRU(in_channels, in_channels))
            if i < n_gru_blocks - 1:
                res.append(nn.Sequential(*[Block(in_channels) for _ in range(n_res_layers)]))
            else:
                res.append(DeepLabHead(in_channels, in_channels, 128))
        self.spatial_grus = nn.ModuleList(grus)
        self.res_blocks = nn.ModuleList(res)

    def forward(self, x, state):
        """x (B,1,latent,H,W): the latent sample; state (B,n_present,C,H,W) -> (B,n_present+n_future,C,H,W).
        Sequences are kept frame-major between the recurrent and the per-frame stages (layers/temporal.py ``stack_frames``)."""
        x = self.dual_grus(x, state)
        x4, restore = frames_as_batch(x)
        x = restore(self.res_blocks1(x4))
        frames = list(unbind_frames(state.to(x.dtype))) + list(unbind_frames(x))
        x = stack_frames(frames)
        hidden = frames[0]
        for gru, res in zip(self.spatial_grus, self.res_blocks):
            x = gru(x, hidden)
            x4, restore = frames_as_batch(x)
            x = restore(res(x4))
        return batch_major(x)
