"""Small host-side helpers."""
import torch
import torch.nn as nn


def hp(x):
    """``x`` in at least float32: bf16 / half are cast up, float32 AND float64 pass through.  Used wherever the host
    code widens an intermediate for accuracy (statistics, per-sample biases, losses): written as ``.float()`` those places
    would round a float64 evaluation of the model -- the noise-free truth the parity tests compare against
    (tests/test_step_truth_cpu.py) -- back to float32."""
    return x if x.dtype in (torch.float32, torch.float64) else x.float()


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


def staged_sum(x, chunk=4096):
    """Sum of all elements as a cascade of row sums, each row reduced inside one workgroup.

    A direct ``x.sum()`` / ``x.mean()`` of a large tensor is a multi-workgroup reduction whose cross-block
    scratch (semaphores) is zeroed with a memset node; replayed from a hipGraph on ROCm 7.2 that scratch was
    observed stale (first replay right, later replays garbage).  The cascade needs no scratch, is
    deterministic, and costs one extra tiny kernel."""
    flat = x.reshape(-1)
    while flat.numel() > 2 * chunk:
        pad = (-flat.numel()) % chunk
        if pad:
            flat = torch.nn.functional.pad(flat, (0, pad))
        flat = flat.view(-1, chunk).sum(1)
    return flat.sum()


def staged_mean(x, chunk=4096):
    return staged_sum(x, chunk) / x.numel()
