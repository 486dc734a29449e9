"""Present-distribution network of the prediction stage: ``stp3/models/distributions.py`` (DistributionModule :7-51,
DistributionEncoder :54-68) -- four stride-2 Bottlenecks, global average pooling and a 1x1 convolution giving the
parameters of a diagonal Gaussian (or a mixture of three, or a per-pixel Bernoulli log-probability)."""
import torch.nn as nn

from ..layers.convolutions import Bottleneck
from ..layers.fused import conv1x1_on_vector, plane_mean


class DistributionEncoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.model = nn.Sequential(Bottleneck(in_channels, out_channels=out_channels, downsample=True),
                                   Bottleneck(out_channels, out_channels=out_channels, downsample=True),
                                   Bottleneck(out_channels, out_channels=out_channels, downsample=True),
                                   Bottleneck(out_channels, out_channels=out_channels, downsample=True))

    def forward(self, s_t):
        return self.model(s_t)


class DistributionModule(nn.Module):
    def __init__(self, in_channels, latent_dim, method='GAUSSIAN'):
        super().__init__()
        self.compress_dim = in_channels // 2
        self.latent_dim = latent_dim
        self.method = method
        if method in ('GAUSSIAN', 'MIXGAUSSIAN'):
            out = 2 * latent_dim if method == 'GAUSSIAN' else 6 * latent_dim + 3
            self.encoder = DistributionEncoder(in_channels, self.compress_dim)
            self.decoder = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(self.compress_dim, out_channels=out, kernel_size=1))
        elif method == 'BERNOULLI':
            self.encoder = nn.Sequential(Bottleneck(in_channels, self.latent_dim))
            self.decoder = nn.LogSigmoid()
        else:
            raise NotImplementedError(method)

    def forward(self, s_t):
        b, s = s_t.shape[:2]
        assert s == 1
        encoding = self.encoder(s_t[:, 0])
        if self.method == 'BERNOULLI':
            return self.decoder(encoding)
        conv = self.decoder[1]
        pooled = plane_mean(encoding).to(conv.weight.dtype)                  # AdaptiveAvgPool2d(1), gradient in x's layout
        out = conv1x1_on_vector(pooled.view(b, -1, 1, 1), conv.weight, conv.bias)
        return out.view(b, 1, -1)
