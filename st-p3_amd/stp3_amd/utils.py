"""Small host-side helpers."""
import torch
import torch.nn as nn


def to_channels_last(module):
    """Put every 2-D / 3-D convolution weight in channels-last memory (what the MI355X conv kernels
    and the pixel-major lift operators want).  ``module.to(memory_format=...)`` cannot be used on
    the whole model because it mixes 4-D and 5-D weights."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        elif isinstance(m, nn.Conv3d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last_3d)
    return module
