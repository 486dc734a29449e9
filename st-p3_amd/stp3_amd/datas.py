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


# ----------------------------------------------------------------------------------------------------------------------
# BEV labels: what NuscenesData.get_birds_eye_view_label / get_label / __getitem__ do AFTER
# the nuScenes devkit has answered (annotation boxes, ego poses): box polygons -> label maps, instance ids -> centerness /
# offset / displacement labels, frames -> one sample dictionary.
# ----------------------------------------------------------------------------------------------------------------------
MAX_POLY_VERTICES = 8


def box_polygons(bottom_corners_xy, bev_start_position, bev_resolution):
    """``_get_poly_region_in_image`` (NuscenesData.py:340-353) after the devkit's ``Box``: the (n, 4, 2) ego-frame bottom
    corners of the annotation boxes -> integer polygon vertices as cv2.fillPoly takes them, (n, 4, 2) int32 with
    [..., 0] the BEV column and [..., 1] the BEV row (the reference rounds to cells, then swaps the two coordinates)."""
    pts = np.asarray(bottom_corners_xy, dtype=np.float64)
    start = np.asarray(bev_start_position, dtype=np.float64)[:2]
    res = np.asarray(bev_resolution, dtype=np.float64)[:2]
    cells = np.round((pts - start + res / 2.0) / res).astype(np.int32)
    return cells[..., [1, 0]]


def fill_polygons_reference(polys, values, map_index, n_maps, hw):
    """cv2.fillPoly, restated (edge walking; the statement CPU tensors take and the tests compare the kernel with):
    see csrc/stp3_labels.hip for the algorithm and its provenance.  polys: sequence of (nv, 2) integer (column, row)
    vertex arrays in paint order -> (n_maps, H, W) float32."""
    h, w = hw
    maps = np.zeros((n_maps, h, w), dtype=np.float32)
    for poly, value, mi in zip(polys, values, map_index):
        poly = np.asarray(poly, dtype=np.int64)
        img = maps[mi]
        nv = len(poly)
        edges = []
        for v in range(nv):
            (ax, ay), (bx, by) = poly[v - 1], poly[v]
            _bresenham(img, int(ax), int(ay), int(bx), int(by), value)
            if ay == by:
                continue
            y0, y1, x0 = (ay, by, ax) if ay < by else (by, ay, bx)
            num = int(bx - ax) << 16
            den = int(by - ay)
            slope = abs(num) // abs(den) * (1 if (num < 0) == (den < 0) else -1)        # C integer division: toward zero
            edges.append((int(y0), int(y1), int(x0) << 16, slope))
        if not edges:
            continue
        for y in range(max(min(e[0] for e in edges), 0), min(max(e[1] for e in edges), h)):
            xs = sorted(x0 + slope * (y - y0) for y0, y1, x0, slope in edges if y0 <= y < y1)
            for left, right in zip(xs[0::2], xs[1::2]):
                x1, x2 = (left + 65535) >> 16, right >> 16
                if x1 < w and x2 >= 0:
                    img[y, max(x1, 0):min(x2, w - 1) + 1] = value
    return torch.from_numpy(maps)


def _bresenham(img, x0, y0, x1, y1, value):
    """OpenCV's LineIterator, 8-connected, walked left to right (drawing.cpp ``Line``), clipped to the image."""
    h, w = img.shape
    if x0 > x1:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, y1 - y0
    sy = -1 if dy < 0 else 1
    ady = abs(dy)
    steep = ady > dx
    big, small = (ady, dx) if steep else (dx, ady)
    err = big - 2 * small
    x, y = x0, y0
    for _ in range(big + 1):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = value
        diag = err < 0
        err += -2 * small + (2 * big if diag else 0)
        if steep:
            y += sy
            x += 1 if diag else 0
        else:
            x += 1
            y += sy if diag else 0


def fill_polygons(polys, values, map_index, n_maps, hw, device=None):
    """cv2.fillPoly of integer polygons into ``n_maps`` zero-initialised (H, W) float32 maps, in order (a later polygon
    overwrites an earlier one); ``polys``: sequence of (nv <= 8, 2) integer (column, row) arrays, ``values`` / ``map_index``
    per polygon.  On a GPU ``device``: one launch of stp3_fill_polygons; otherwise the restatement above."""
    device = torch.device(device) if device is not None else torch.device('cpu')
    if not torch.empty(0, device=device).is_cuda:
        return fill_polygons_reference(polys, values, map_index, n_maps, hw)
    h, w = hw
    arr = (_lib.Poly * max(len(polys), 1))()
    for rec, poly, value, mi in zip(arr, polys, values, map_index):
        poly = np.asarray(poly, dtype=np.int64)
        if not 1 <= len(poly) <= MAX_POLY_VERTICES:
            raise _lib.Stp3HipError(f'fill_polygons: {len(poly)} vertices (1 .. {MAX_POLY_VERTICES} supported)')
        rec.map, rec.nv, rec.value = int(mi), len(poly), float(value)
        for k, (px, py) in enumerate(poly):
            rec.xy[2 * k], rec.xy[2 * k + 1] = int(px), int(py)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    maps = torch.zeros(n_maps, h, w, dtype=torch.float32, device=device)
    _lib.check(_lib.lib().stp3_fill_polygons(ops._ptr(table), len(polys), n_maps, h, w, ops._ptr(maps), ops._stream()),
               'stp3_fill_polygons')
    return maps


def bev_labels_from_boxes(bottom_corners_xy, categories, instance_ids, bev_start_position, bev_resolution, bev_dimension,
                          device=None):
    """``get_birds_eye_view_label`` (NuscenesData.py:303-338) for one frame, after the devkit: boxes (n, 4, 2) ego-frame
    bottom corners in annotation order, ``categories`` 'vehicle' / 'human' / anything else (skipped), ``instance_ids`` the
    ids the reference's ``instance_map`` assigns -> (segmentation, instance, pedestrian) int64 maps (H, W).  Vehicles paint
    their id into the instance map (later boxes overwrite earlier ones) and 1 into the segmentation, pedestrians 1 into
    the pedestrian map."""
    h, w = int(bev_dimension[0]), int(bev_dimension[1])
    polys = box_polygons(bottom_corners_xy, bev_start_position, bev_resolution) if len(categories) else []
    plist, values, index = [], [], []
    for poly, cat, iid in zip(polys, categories, instance_ids):
        if 'vehicle' in cat:
            plist += [poly, poly]
            values += [float(iid), 1.0]
            index += [1, 0]
        elif 'human' in cat:
            plist.append(poly)
            values.append(1.0)
            index.append(2)
    maps = fill_polygons(plist, values, index, 3, (h, w), device).long()
    return maps[0], maps[1], maps[2]


def instance_labels_reference(instance, future_egomotion, num_instances, ignore_index=255, subtract_egomotion=True,
                              sigma=3, spatial_extent=None):
    """``convert_instance_mask_to_center_and_offset_label`` (stp3/utils/instance.py:12-77) with whole-tensor torch
    operators (CPU tensors; the same arithmetic as the kernel: integer moments, float32 mean, round half to even)."""
    from .geometry import mat2pose_vec, pose_vec2mat, warp_features
    t_, h, w = instance.shape
    dev = instance.device
    center = torch.zeros(t_, 1, h, w, device=dev)
    offset = ignore_index * torch.ones(t_, 2, h, w, device=dev)
    flow = ignore_index * torch.ones(t_, 2, h, w, device=dev)
    if num_instances == 0:
        return center, offset, flow
    warped = _warp_instances(instance, future_egomotion, subtract_egomotion, spatial_extent)
    ids = torch.arange(1, num_instances + 1, device=dev).view(1, -1, 1, 1)
    rows = torch.arange(h, dtype=torch.float32, device=dev).view(1, 1, h, 1)
    cols = torch.arange(w, dtype=torch.float32, device=dev).view(1, 1, 1, w)

    def moments(maps):                                  # (T, K): count, rounded mean row, rounded mean column
        m = (maps.unsqueeze(1) == ids).float()
        cnt = m.sum(dim=(2, 3))
        safe = cnt.clamp_min(1.0)
        return cnt, torch.round((m * rows).sum(dim=(2, 3)) / safe), torch.round((m * cols).sum(dim=(2, 3)) / safe)
    cnt, xc, yc = moments(instance)
    wcnt, wxc, wyc = moments(warped)
    present = cnt > 0
    ox = xc.view(t_, -1, 1, 1) - rows                   # (T, K, H, W)
    oy = yc.view(t_, -1, 1, 1) - cols
    g = torch.exp(-(ox ** 2 + oy ** 2) / sigma ** 2)
    center[:, 0] = torch.where(present.view(t_, -1, 1, 1), g, torch.zeros(())).amax(dim=1)
    own = (instance.unsqueeze(1) == ids)                # (T, K, H, W): at most one id per pixel
    sel = own.float()
    has = own.any(dim=1)
    offset[:, 0] = torch.where(has, (sel * ox).sum(1), offset[:, 0])
    offset[:, 1] = torch.where(has, (sel * oy).sum(1), offset[:, 1])
    if t_ > 1:
        ok = present[:-1] & present[1:] & (wcnt[1:] > 0)                     # (T-1, K)
        dxy = torch.stack([wxc[1:] - xc[:-1], wyc[1:] - yc[:-1]], dim=1)     # (T-1, 2, K)
        moving = own[:-1] & ok.view(t_ - 1, -1, 1, 1)
        hit = moving.any(dim=1)
        for a in range(2):
            val = (moving.float() * dxy[:, a].view(t_ - 1, -1, 1, 1)).sum(1)
            flow[:-1, a] = torch.where(hit, val, flow[:-1, a])
    return center, offset, flow


def _warp_instances(instance, future_egomotion, subtract_egomotion, spatial_extent):
    """Frame t's instance map warped into frame t - 1 (instance.py:21-31); frame 0 is a copy that nothing reads.  The
    reference applies the warp only when ``subtract_egomotion`` (its inverse-motion tensor does not exist otherwise)."""
    from .geometry import mat2pose_vec, pose_vec2mat, warp_features
    if not subtract_egomotion:
        raise NotImplementedError('subtract_egomotion=False: the reference itself fails (future_egomotion_inv is undefined)')
    inv = mat2pose_vec(torch.inverse(pose_vec2mat(future_egomotion.float())))
    out = instance.float().clone()
    if instance.shape[0] > 1:
        out[1:] = warp_features(instance[1:].unsqueeze(1).float(), inv[:-1].to(instance.device), mode='nearest',
                                spatial_extent=spatial_extent)[:, 0]
    return out


def instance_labels(instance, future_egomotion, num_instances, ignore_index=255, subtract_egomotion=True, sigma=3,
                    spatial_extent=None):
    """``convert_instance_mask_to_center_and_offset_label`` (stp3/utils/instance.py:12-77): instance (T, H, W) int64 ids,
    future_egomotion (T, 6) -> centerness (T, 1, H, W), offset (T, 2, H, W), future displacement (T, 2, H, W), float32.
    GPU tensors: the warp through stp3_warp_nearest, the labels through stp3_instance_labels (integer moments, no
    per-instance Python loop)."""
    if not instance.is_cuda:
        return instance_labels_reference(instance, future_egomotion, num_instances, ignore_index, subtract_egomotion, sigma,
                                         spatial_extent)
    from . import ops_loss
    from .geometry import mat2pose_vec, pose_vec2mat, warp_theta
    t_, h, w = instance.shape
    dev = instance.device
    inst = instance.long().contiguous()
    warped = None
    if t_ > 1:
        if not subtract_egomotion:
            raise NotImplementedError('subtract_egomotion=False: the reference itself fails (future_egomotion_inv is undefined)')
        inv = mat2pose_vec(torch.inverse(pose_vec2mat(future_egomotion.detach().float().cpu())))
        theta = torch.cat([torch.zeros(1, 2, 3), warp_theta(inv[:-1], spatial_extent)], dim=0)
        warped = ops_loss.warp_nearest(inst.float().unsqueeze(1), theta, [1] + [0] * (t_ - 1))[:, 0].contiguous()
    lib = _lib.lib()
    need = ctypes.c_size_t()
    _lib.check(lib.stp3_instance_labels_workspace_bytes(t_, int(num_instances), ctypes.byref(need)),
               'stp3_instance_labels_workspace_bytes')
    ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device=dev)
    center = torch.empty(t_, 1, h, w, dtype=torch.float32, device=dev)
    offset = torch.empty(t_, 2, h, w, dtype=torch.float32, device=dev)
    flow = torch.empty(t_, 2, h, w, dtype=torch.float32, device=dev)
    _lib.check(lib.stp3_instance_labels(t_, h, w, int(num_instances), float(ignore_index), float(sigma), ops._ptr(inst),
                                        ops._ptr(warped) if warped is not None else None, ops._ptr(ws), need.value,
                                        ops._ptr(center), ops._ptr(offset), ops._ptr(flow), ops._stream()),
               'stp3_instance_labels')
    return center, offset, flow


SAMPLE_CAT_KEYS = ('image', 'intrinsics', 'extrinsics', 'depths', 'segmentation', 'instance', 'future_egomotion', 'hdmap',
                   'pedestrian')


def assemble_sample(frames, receptive_field, num_instances, planning=None, gt_depth=False, ignore_index=255,
                    spatial_extent=None):
    """``NuscenesData.__getitem__`` (NuscenesData.py:569-646) after the per-frame queries: ``frames`` = one dictionary per
    time step with the tensors the reference's helpers return for it -- image (1, N, 3, H, W), intrinsics (1, N, 3, 3),
    extrinsics (1, N, 4, 4) [, depths (1, N, H, W)] for the first ``receptive_field`` frames; segmentation / pedestrian
    (1, 1, X, Y), instance (1, X, Y), future_egomotion (1, 6), hdmap (1, C, X, Y) and ``index`` for all -- concatenated
    over time under the reference's keys, plus the instance-derived labels.  ``planning``: the present frame's
    gt_trajectory / command / sample_trajectory (dict), placeholders of the reference's shapes otherwise."""
    data = {k: [] for k in SAMPLE_CAT_KEYS}
    data['indices'] = []
    for i, fr in enumerate(frames):
        if i < receptive_field:
            for k in ('image', 'intrinsics', 'extrinsics'):
                data[k].append(fr[k])
            if gt_depth:
                data['depths'].append(fr['depths'])
        for k in ('segmentation', 'instance', 'pedestrian', 'future_egomotion', 'hdmap'):
            data[k].append(fr[k])
        data['indices'].append(fr.get('index', i))
    for k in SAMPLE_CAT_KEYS:
        if k == 'depths' and not gt_depth:
            continue
        data[k] = torch.cat(data[k], dim=0)
    planning = planning or {}
    data['gt_trajectory'] = planning.get('gt_trajectory', torch.zeros(1, 3))
    data['command'] = planning.get('command', 'FORWARD')
    data['sample_trajectory'] = planning.get('sample_trajectory', torch.zeros(1, 1, 3))
    data['target_point'] = torch.tensor([0., 0.])
    data['centerness'], data['offset'], data['flow'] = instance_labels(
        data['instance'], data['future_egomotion'], num_instances, ignore_index=ignore_index, subtract_egomotion=True,
        spatial_extent=spatial_extent)
    return data
