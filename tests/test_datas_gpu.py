"""GPU: the image path on the MI355X -- stp3_image_prep against the fixtures made with the
reference's ``resize_and_crop_image`` + Pillow (tests/golden/image_prep.npz), byte for byte; at nuScenes size (900 x
1600 -> 224 x 480) through the sha256 of the resized bytes; against Pillow directly when it is importable; bf16 output =
the float32 output rounded once."""
import hashlib

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_datas_cpu import CASES, as_bytes, preprocessor

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(CASES))
def test_kernel_matches_pillow_fixture(name):
    g = H.load('image_prep.npz')
    prep, images = preprocessor(name)
    y = prep(images.cuda())
    assert y.is_cuda and np.array_equal(as_bytes(y), g[f'{name}/bytes'])
    np.testing.assert_allclose(y.cpu().numpy(), g[f'{name}/normalised'], rtol=0, atol=2.4e-7)
    assert torch.equal(prep(images.cuda(), out_dtype=torch.bfloat16), y.to(torch.bfloat16))


def test_nuscenes_size_bytes_and_batch_layout():
    from stp3_amd.config import perception_cfg
    from stp3_amd.datas import ImagePreprocessor
    g = H.load('image_prep.npz')
    prep = ImagePreprocessor(perception_cfg())
    images = torch.from_numpy(H.image_bytes((2, 900, 1600, 3), 402)).cuda()
    y = prep(images)
    assert tuple(y.shape) == (2, 3, 224, 480)
    assert hashlib.sha256(as_bytes(y).tobytes()).digest() == g['nuscenes/sha256'].tobytes()
    np.testing.assert_allclose(y.cpu().numpy().reshape(-1)[::97], g['nuscenes/normalised_sample'], rtol=0, atol=2.4e-7)
    # leading dimensions are kept: (B, S, N, H, W, 3) -> (B, S, N, 3, h, w), the model's input
    stacked = images.view(1, 1, 2, 900, 1600, 3)
    assert torch.equal(prep(stacked), y.view(1, 1, 2, 3, 224, 480))
    assert torch.equal(prep(images), prep.reference(images.cpu()).cuda())        # the torch statements, same floats


def test_against_pillow_directly():
    pytest.importorskip('PIL')
    from PIL import Image
    from stp3_amd.datas import ImagePreprocessor
    for i, (h, w, wr, hr) in enumerate(((37, 53, 20, 30), (64, 64, 100, 90), (450, 800, 240, 135))):
        a = H.image_bytes((1, h, w, 3), 430 + i)
        want = np.asarray(Image.fromarray(a[0]).resize((wr, hr), resample=Image.BILINEAR))
        prep = ImagePreprocessor(resize_dims=(wr, hr), crop=(0, 0, wr, hr), source_hw=(h, w))
        assert np.array_equal(as_bytes(prep(torch.from_numpy(a).cuda()))[0], want), (h, w, wr, hr)
