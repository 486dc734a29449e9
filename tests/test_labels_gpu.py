"""GPU: csrc/stp3_labels.hip -- stp3_fill_polygons and stp3_instance_labels on the MI355X against tests/golden/labels.npz
(the reference's own instance-label function; the oracle's restatement of cv2.fillPoly): see tests/test_labels_cpu.py."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_labels_cpu import BEV_DIM, BEV_RES, BEV_START, check_instance_labels, fixture

pytestmark = pytest.mark.gpu


def test_instance_labels_kernel_matches_the_reference():
    check_instance_labels('cuda')


def test_polygon_fill_kernel():
    from stp3_amd import datas
    g = fixture()
    polys = list(g['poly/vertices'])
    want = np.unpackbits(g['poly/oracle'], axis=1).reshape(40, 200, 200)
    got = datas.fill_polygons(polys, [1.0] * 40, list(range(40)), 40, (200, 200), device='cuda')
    assert got.is_cuda and np.array_equal(got.cpu().numpy().astype(np.uint8), want)
    # all of them into ONE map with their index as value: the paint order decides the overlaps
    one = datas.fill_polygons(polys, [float(i + 1) for i in range(40)], [0] * 40, 1, (200, 200), device='cuda').cpu().numpy()[0]
    ref = datas.fill_polygons(polys, [float(i + 1) for i in range(40)], [0] * 40, 1, (200, 200)).numpy()[0]
    assert np.array_equal(one, ref)
    # random polygons with 3 .. 8 vertices (not necessarily convex), partly outside the image: kernel == CPU statement
    rng = np.random.default_rng(5)
    many = [rng.integers(-20, 120, (int(rng.integers(3, 9)), 2)) for _ in range(60)]
    a = datas.fill_polygons(many, [1.0] * 60, list(range(60)), 60, (96, 104), device='cuda').cpu().numpy()
    b = datas.fill_polygons(many, [1.0] * 60, list(range(60)), 60, (96, 104)).numpy()
    assert np.array_equal(a, b)
    # thin, degenerate and self-touching polygons on a tiny lattice: edges meet exactly at pixel centres all the time
    thin = [rng.integers(0, 14, (int(rng.integers(3, 9)), 2)) for _ in range(300)]
    c = datas.fill_polygons(thin, [1.0] * 300, list(range(300)), 300, (16, 16), device='cuda').cpu().numpy()
    d = datas.fill_polygons(thin, [1.0] * 300, list(range(300)), 300, (16, 16)).numpy()
    assert np.array_equal(c, d)


def test_label_maps_from_boxes_on_the_gpu():
    from stp3_amd import datas
    g = fixture()
    corners = g['poly/corners']
    cats = ['vehicle.car' if i % 3 else 'human.pedestrian.adult' for i in range(len(corners))]
    ids = list(range(1, len(corners) + 1))
    on_gpu = datas.bev_labels_from_boxes(corners, cats, ids, BEV_START, BEV_RES, BEV_DIM, device='cuda')
    on_cpu = datas.bev_labels_from_boxes(corners, cats, ids, BEV_START, BEV_RES, BEV_DIM)
    for a, b in zip(on_gpu, on_cpu):
        assert a.is_cuda and torch.equal(a.cpu(), b)
