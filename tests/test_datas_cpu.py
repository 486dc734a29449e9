"""CPU: the image path against fixtures produced by the reference's own
``resize_and_crop_image`` with Pillow (oracle/make_golden_image.py -> tests/golden/image_prep.npz): the torch statements
of ``stp3_amd.datas`` (the route CPU tensors take) reproduce Pillow's resized BYTES exactly and the normalised floats
to float32 rounding; the intrinsics update; the augmentation parameters of the default configuration.  The HIP kernel
behind the same call is checked on the CPU stand-in (tests/test_kernels_on_cpu.py, case `image_prep`) and on the MI355X
(tests/test_datas_gpu.py)."""
import hashlib

import numpy as np
import pytest
import torch

from tests import helpers as H

CASES = {'small': ((3, 90, 160, 3), 401, (48, 27), (2, 5, 46, 25)),
         'padded': ((3, 90, 160, 3), 401, (48, 27), (-3, 5, 51, 30))}


def preprocessor(name):
    from stp3_amd.datas import ImagePreprocessor
    shape, seed, resize_dims, crop = CASES[name]
    return ImagePreprocessor(resize_dims=resize_dims, crop=crop, source_hw=shape[1:3]), torch.from_numpy(H.image_bytes(shape, seed))


def as_bytes(normalised):
    """Invert ToTensor + Normalize: the resized bytes behind a normalised tensor (N, 3, h, w) -> (N, h, w, 3) uint8."""
    from stp3_amd.datas import IMAGENET_MEAN, IMAGENET_STD
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return ((normalised.float().cpu() * std + mean) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize('name', sorted(CASES))
def test_torch_statements_match_pillow(name):
    g = H.load('image_prep.npz')
    prep, images = preprocessor(name)
    y = prep(images)
    assert np.array_equal(as_bytes(y), g[f'{name}/bytes'])
    np.testing.assert_allclose(y.numpy(), g[f'{name}/normalised'], rtol=0, atol=2.4e-7)      # one float32 rounding


def test_resize_matches_installed_pillow_directly():
    """Pillow is installed in this image: up- and down-scaling, both axes, odd sizes."""
    pytest.importorskip('PIL')
    from PIL import Image
    from stp3_amd.datas import resize_bilinear_pil
    for i, (h, w, wr, hr) in enumerate(((37, 53, 20, 30), (64, 64, 100, 90), (90, 160, 48, 27), (11, 200, 60, 11))):
        a = H.image_bytes((h, w, 3), 410 + i)
        want = np.asarray(Image.fromarray(a).resize((wr, hr), resample=Image.BILINEAR))
        got = resize_bilinear_pil(torch.from_numpy(a), (wr, hr)).numpy()
        assert np.array_equal(got, want), (h, w, wr, hr)


def test_default_configuration_parameters_and_intrinsics():
    from stp3_amd.config import perception_cfg
    from stp3_amd.datas import ImagePreprocessor, get_resizing_and_cropping_parameters
    cfg = perception_cfg()
    p = get_resizing_and_cropping_parameters(cfg)
    assert p['resize_dims'] == (480, 270) and p['crop'] == (0, 46, 480, 270)               # NuscenesData.py:150-172
    prep = ImagePreprocessor(cfg)
    assert prep.kk_h.shape == (480, 9) and prep.kk_v.shape == (270, 9)                     # scale 10/3: support 3.33
    k = torch.tensor([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    np.testing.assert_array_equal(prep.intrinsics(k).numpy(), H.load('image_prep.npz')['intrinsics'])


def test_nuscenes_size_matches_pillow():
    from stp3_amd.config import perception_cfg
    from stp3_amd.datas import ImagePreprocessor
    g = H.load('image_prep.npz')
    y = ImagePreprocessor(perception_cfg())(torch.from_numpy(H.image_bytes((2, 900, 1600, 3), 402)))
    assert hashlib.sha256(as_bytes(y).tobytes()).digest() == g['nuscenes/sha256'].tobytes()
