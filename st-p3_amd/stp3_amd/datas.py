"""Input side of the path: what the reference's loader does to every camera image between
the decoder and the network (``stp3/datas/NuscenesData.py:150-172, 236-253``; ``stp3/utils/geometry.py:9-37``) --

    PIL resize (BILINEAR) to ``resize_dims`` -> crop -> ToTensor (/ 255) -> Normalize(ImageNet mean / std)
    and the matching update of the camera intrinsics

-- for all images of a batch in ONE launch of ``stp3_image_prep`` (csrc/stp3_image.hip), byte-exact with Pillow's
resampler.  The nuScenes devkit queries around it (sample records, CAN bus, map API, annotation boxes) are third-party
data access and stay what they are.

CPU tensors take ``resize_bilinear_pil`` / ``ImagePreprocessor.reference`` below -- the same integer arithmetic written
with torch operators -- which is what the host-side tests pin against Pillow itself.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib, ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)              # NuscenesData.py:68-72
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2                         # Pillow, src/libImaging/Resample.c


def get_resizing_and_cropping_parameters(cfg):
    """``NuscenesData.get_resizing_and_cropping_parameters`` (:150-172): resize by IMAGE.RESIZE_SCALE, crop TOP_CROP rows
    and centre the FINAL_DIM window horizontally."""
    oh, ow = cfg.IMAGE.ORIGINAL_HEIGHT, cfg.IMAGE.ORIGINAL_WIDTH
    fh, fw = cfg.IMAGE.FINAL_DIM
    scale = cfg.IMAGE.RESIZE_SCALE
    resize_dims = (int(ow * scale), int(oh * scale))
    crop_h = cfg.IMAGE.TOP_CROP
    crop_w = int(max(0, (resize_dims[0] - fw) / 2))
    return {'scale_width': scale, 'scale_height': scale, 'resize_dims': resize_dims,
            'crop': (crop_w, crop_h, crop_w + fw, crop_h + fh)}


def update_intrinsics(intrinsics, top_crop=0.0, left_crop=0.0, scale_width=1.0, scale_height=1.0):
    """Focal lengths and principal point after resize + crop (stp3/utils/geometry.py:16-37); (..., 3, 3)."""
    k = intrinsics.clone()
    k[..., 0, 0] *= scale_width
    k[..., 0, 2] *= scale_width
    k[..., 1, 1] *= scale_height
    k[..., 1, 2] *= scale_height
    k[..., 0, 2] -= left_crop
    k[..., 1, 2] -= top_crop
    return k


def pil_bilinear_coefficients(in_size, out_size):
    """Pillow's resampling coefficients for one axis (Resample.c ``precompute_coeffs`` with the bilinear (triangle)
    filter over the whole axis, then ``normalize_coeffs_8bpc``): (kk (out_size, ksize) int32 in 22-bit fixed point,
    bounds (out_size, 2) int32 = first source index and count).  Python floats are C doubles: the same numbers."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[xx, x] = int(v * one - 0.5) if v < 0 else int(v * one + 0.5)
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _resample_axis(img, kk, bounds, axis):
    """One Pillow pass along ``axis`` of an integer tensor of bytes: sum of taps in fixed point, + 2^21, >> 22, clip."""
    src = img.movedim(axis, 0).to(torch.int64)
    kk_t, b_t = torch.from_numpy(kk).to(img.device).long(), torch.from_numpy(bounds).to(img.device).long()
    taps = torch.arange(kk.shape[1], device=img.device)
    idx = (b_t[:, :1] + taps).clamp(max=src.shape[0] - 1)                  # (out, ksize); weights beyond the count are 0
    w = torch.where(taps[None] < b_t[:, 1:], kk_t, torch.zeros_like(kk_t))
    g = src[idx]                                                            # (out, ksize, ...)
    acc = (g * w.view(*w.shape, *([1] * (src.dim() - 1)))).sum(dim=1) + (1 << (PRECISION_BITS - 1))
    return (acc >> PRECISION_BITS).clamp(0, 255).to(torch.uint8).movedim(0, axis)


def resize_bilinear_pil(images, resize_dims):
    """``PIL.Image.resize(resize_dims, BILINEAR)`` of (..., H, W, 3) uint8 images, bit for bit (horizontal pass, bytes,
    vertical pass -- Pillow's order), with torch operators."""
    wr, hr = resize_dims
    h, w = images.shape[-3], images.shape[-2]
    kh, bh = pil_bilinear_coefficients(w, wr)
    kv, bv = pil_bilinear_coefficients(h, hr)
    return _resample_axis(_resample_axis(images, kh, bh, images.dim() - 2), kv, bv, images.dim() - 3)


class ImagePreprocessor:
    """``resize_and_crop_image`` + ``normalise_image`` of the reference's loader for whole batches.

        prep = ImagePreprocessor(cfg)
        x = prep(images_uint8)            # (..., H, W, 3) uint8 on the GPU -> (..., 3, FINAL_H, FINAL_W) float32 | bf16
        k = prep.intrinsics(k_raw)        # (..., 3, 3)
    """

    def __init__(self, cfg=None, resize_dims=None, crop=None, source_hw=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        if cfg is not None:
            p = get_resizing_and_cropping_parameters(cfg)
            resize_dims, crop = p['resize_dims'], p['crop']
            source_hw = (cfg.IMAGE.ORIGINAL_HEIGHT, cfg.IMAGE.ORIGINAL_WIDTH)
            self.scale = (p['scale_width'], p['scale_height'])
        else:
            self.scale = (resize_dims[0] / source_hw[1], resize_dims[1] / source_hw[0])
        self.resize_dims, self.crop, self.source_hw = tuple(resize_dims), tuple(crop), tuple(source_hw)
        self.mean, self.std = tuple(mean), tuple(std)
        self.kk_h, self.bounds_h = pil_bilinear_coefficients(source_hw[1], resize_dims[0])
        self.kk_v, self.bounds_v = pil_bilinear_coefficients(source_hw[0], resize_dims[1])
        self._tables = {}

    def intrinsics(self, intrinsics):
        return update_intrinsics(intrinsics, self.crop[1], self.crop[0], self.scale[0], self.scale[1])

    def reference(self, images):
        """The torch statements of the same chain (CPU tensors; what the tests compare with Pillow)."""
        left, top, right, bottom = self.crop
        wr, hr = self.resize_dims
        small = resize_bilinear_pil(images, self.resize_dims)
        window = torch.zeros(*images.shape[:-3], bottom - top, right - left, 3, dtype=torch.uint8, device=images.device)
        y0, y1, x0, x1 = max(top, 0), min(bottom, hr), max(left, 0), min(right, wr)      # PIL pads a crop with zeros
        window[..., y0 - top:y1 - top, x0 - left:x1 - left, :] = small[..., y0:y1, x0:x1, :]
        x = window.movedim(-1, -3).to(torch.float32).div(255)                               # ToTensor
        mean = torch.tensor(self.mean, dtype=torch.float32, device=x.device).view(3, 1, 1)
        std = torch.tensor(self.std, dtype=torch.float32, device=x.device).view(3, 1, 1)
        return (x - mean) / std                                                              # Normalize

    def _strip_rows(self, rows_per_wg):
        left, top, right, bottom = self.crop
        hr = self.resize_dims[1]
        worst = 1
        for r0 in range(0, bottom - top, rows_per_wg):
            rows = [y for y in range(top + r0, min(top + r0 + rows_per_wg, bottom)) if 0 <= y < hr]
            if rows:
                lo = min(int(self.bounds_v[y, 0]) for y in rows)
                hi = max(int(self.bounds_v[y, 0] + self.bounds_v[y, 1]) for y in rows)
                worst = max(worst, hi - lo)
        return worst

    def __call__(self, images, out_dtype=torch.float32):
        if not images.is_cuda:
            return self.reference(images).to(out_dtype)
        assert images.dtype == torch.uint8 and images.shape[-1] == 3 and tuple(images.shape[-3:-1]) == self.source_hw
        lead = images.shape[:-3]
        flat = images.reshape(-1, *images.shape[-3:]).contiguous()
        dev = flat.device
        lib = _lib.lib()
        key = str(dev)
        if key not in self._tables:
            self._tables[key] = tuple(torch.from_numpy(a).to(dev).contiguous()
                                      for a in (self.kk_h, self.bounds_h, self.kk_v, self.bounds_v)) + (
                                          self._strip_rows(lib.stp3_image_prep_rows_per_workgroup()),)
        kk_h, b_h, kk_v, b_v, strip = self._tables[key]
        left, top, right, bottom = self.crop
        d = _lib.ImageDims()
        d.N, d.H, d.W = flat.shape[0], self.source_hw[0], self.source_hw[1]
        d.Wr, d.Hr = self.resize_dims
        d.left, d.top, d.Wo, d.Ho = left, top, right - left, bottom - top
        d.ksize_h, d.ksize_v = self.kk_h.shape[1], self.kk_v.shape[1]
        d.out_dtype = _lib.DTYPE_BF16 if out_dtype == torch.bfloat16 else _lib.DTYPE_F32
        for c in range(3):
            d.mean[c], d.std[c] = self.mean[c], self.std[c]
        out = torch.empty(flat.shape[0], 3, d.Ho, d.Wo, dtype=out_dtype, device=dev)
        _lib.check(lib.stp3_image_prep(ctypes.byref(d), ops._ptr(flat), ops._ptr(kk_h), ops._ptr(b_h), ops._ptr(kk_v),
                                       ops._ptr(b_v), strip, ops._ptr(out), ops._stream()), 'stp3_image_prep')
        return out.view(*lead, 3, d.Ho, d.Wo)
