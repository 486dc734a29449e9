"""stp3_image_prep on the MI355X: one batch of camera images (B x 3 frames x 6 cameras, 900 x 1600 x 3 bytes each) to
network input, float32 and bf16: microseconds, bytes moved (source read once + output written once) against 8 TB/s; and
the same chain as the reference's loader runs it -- per image with Pillow + torch on the host cores -- when Pillow is
importable.

    python scripts/time_image.py [batch]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
sys.path.insert(0, ROOT)
from stp3_amd.config import perception_cfg  # noqa: E402
from stp3_amd.datas import IMAGENET_MEAN, IMAGENET_STD, ImagePreprocessor  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = B * 3 * 6
    prep = ImagePreprocessor(perception_cfg())
    g = torch.Generator().manual_seed(0)
    images = torch.randint(0, 256, (n, 900, 1600, 3), dtype=torch.uint8, generator=g).cuda()
    for dtype in (torch.float32, torch.bfloat16):
        for _ in range(3):
            prep(images, out_dtype=dtype)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            y = prep(images, out_dtype=dtype)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        moved = images.numel() + y.numel() * y.element_size()
        print(f'{n} images -> {tuple(y.shape)} {str(dtype)[6:]}: {us:8.1f} us, {moved / 1e6:.0f} MB, '
              f'{moved / us / 1e3:.0f} GB/s = {moved / us / 1e3 / 8000:.3f} of 8 TB/s; {n / us * 1e6:.0f} images/s')
    try:
        from PIL import Image
    except ImportError:
        return
    host = images[:6].cpu().numpy()
    mean, std = torch.tensor(IMAGENET_MEAN).view(3, 1, 1), torch.tensor(IMAGENET_STD).view(3, 1, 1)
    t0 = time.perf_counter()
    for a in host:
        img = Image.fromarray(a).resize((480, 270), resample=Image.BILINEAR).crop((0, 46, 480, 270))
        t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)
        t = (t - mean) / std
    dt = (time.perf_counter() - t0) / len(host)
    print(f'reference chain on one host core (Pillow resize + crop, ToTensor, Normalize): {dt * 1e3:.2f} ms per image '
          f'= {1 / dt:.0f} images/s')


if __name__ == '__main__':
    main()
