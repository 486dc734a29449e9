"""CPU: the BEV label path -- box polygons -> cv2.fillPoly maps, instance ids -> centerness /
offset / displacement labels, frames -> one sample -- against tests/golden/labels.npz (oracle/make_golden_labels.py):

  * instance labels: the reference's OWN ``convert_instance_mask_to_center_and_offset_label`` (first-party, pinned):
    offsets and displacements exactly, centerness to one float32 rounding of exp;
  * polygons: cv2 is not installed (parity unpinned, stated): three formulations of OpenCV's published algorithm agree
    pixel for pixel -- the oracle (brute force per pixel), the product's CPU statement (edge walking) and, on the GPU /
    the stand-in, the kernel (closed form per pixel) -- and contain what Pillow's ImageDraw.polygon paints for the same
    vertices (Pillow's outline is thinner: the fixture records 8 % fewer pixels, all of them on the boundary);
  * ``assemble_sample``: keys, shapes and dtypes of NuscenesData.__getitem__ (stp3/datas/NuscenesData.py:569-646)."""
import numpy as np
import torch

from tests import helpers as H

BEV_START, BEV_RES, BEV_DIM = (-49.75, -49.75, 0.0), (0.5, 0.5, 20.0), (200, 200, 1)


def fixture():
    return H.load('labels.npz')


def check_instance_labels(device):
    from stp3_amd import datas
    g = fixture()
    inst = torch.from_numpy(g['instance/ids'].astype(np.int64)).to(device)
    ego = torch.from_numpy(g['instance/future_egomotion'])
    k = int(g['instance/num_instances'][0])
    center, offset, flow = datas.instance_labels(inst, ego.to(device), k, ignore_index=255, subtract_egomotion=True,
                                                 spatial_extent=(50.0, 50.0))
    assert center.shape == (5, 1, 200, 200) and offset.shape == (5, 2, 200, 200) and flow.shape == (5, 2, 200, 200)
    assert torch.equal(offset.cpu(), torch.from_numpy(g['instance/offset']))
    assert torch.equal(flow.cpu(), torch.from_numpy(g['instance/flow']))
    torch.testing.assert_close(center.cpu(), torch.from_numpy(g['instance/center']), rtol=2e-6, atol=1e-7)
    # the fixture exercises every branch: displacement labels exist, and some instance pixels have none
    f, o = torch.from_numpy(g['instance/flow']), torch.from_numpy(g['instance/offset'])
    assert (f != 255).any() and ((o[:-1] != 255) & (f[:-1] == 255)).any()


def test_instance_labels_match_the_reference():
    check_instance_labels('cpu')


def test_polygon_fill_three_ways():
    from oracle import labels_oracle as lo
    from stp3_amd import datas
    g = fixture()
    polys = datas.box_polygons(g['poly/corners'], BEV_START, BEV_RES)
    assert np.array_equal(polys, g['poly/vertices'])
    want = np.unpackbits(g['poly/oracle'], axis=1).reshape(40, 200, 200)
    got = datas.fill_polygons(list(polys), [1.0] * 40, list(range(40)), 40, (200, 200)).numpy()
    assert np.array_equal(got.astype(np.uint8), want)
    # a few of them through the brute-force oracle again (the fixture came from it: guards against a stale fixture)
    for i in (0, 7, 23):
        assert np.array_equal(lo.fill_poly(np.zeros((200, 200), np.float32), polys[i], 1.0).astype(np.uint8), want[i])
    # Pillow (independent, installed): everything it paints is painted here too; the rest lies on the outline
    pil = np.unpackbits(g['poly/pillow'], axis=1).reshape(40, 200, 200)
    assert not (pil > want).any()
    assert (want > pil).sum() <= 0.1 * want.sum()
    # degenerate and clipped cases do not fall over: a point, a segment, a polygon outside the image
    odd = datas.fill_polygons([np.array([[5, 5]]), np.array([[1, 1], [8, 3]]), np.array([[-30, -30], [-20, -30], [-20, -20]]),
                               np.array([[195, 190], [210, 190], [210, 205], [195, 205]])], [1, 2, 3, 4], [0, 0, 0, 0], 1, (200, 200)).numpy()[0]
    assert odd[5, 5] == 1 and odd[1, 1] == 2 and odd[3, 8] == 2 and (odd == 3).sum() == 0 and odd[199, 199] == 4 and odd[189, 199] == 0


def test_later_boxes_overwrite_earlier_ones_and_label_maps():
    from stp3_amd import datas
    corners = np.array([[[0, 0], [4, 0], [4, 2], [0, 2]], [[2, 0], [6, 0], [6, 2], [2, 2]], [[10, 10], [11, 10], [11, 11], [10, 11]]],
                       dtype=np.float64)
    seg, inst, ped = datas.bev_labels_from_boxes(corners, ['vehicle.car', 'vehicle.truck', 'human.pedestrian.adult'], [1, 2, 3],
                                                 BEV_START, BEV_RES, BEV_DIM)
    assert seg.dtype == torch.int64 and seg.shape == (200, 200)
    assert set(inst.unique().tolist()) == {0, 1, 2} and ped.sum() > 0 and (seg > 0).sum() == (inst > 0).sum()
    # box 2 was painted after box 1: the cells they share carry id 2 (cv2.fillPoly paints over)
    rows, cols = (inst == 2).nonzero(as_tuple=True)
    one = (inst == 1).nonzero(as_tuple=True)
    assert cols.min() > one[1].min() or rows.min() > one[0].min()
    shared = datas.bev_labels_from_boxes(corners[:2], ['vehicle.car', 'vehicle.truck'], [2, 1], BEV_START, BEV_RES, BEV_DIM)[1]
    assert (shared == 1).sum() == (inst == 2).sum()                   # same geometry, ids exchanged


def test_assemble_sample_has_the_loaders_schema():
    from stp3_amd import datas
    t_total, rf, n = 5, 3, 6
    frames = []
    for t in range(t_total):
        fr = {'segmentation': torch.zeros(1, 1, 200, 200, dtype=torch.int64), 'pedestrian': torch.zeros(1, 1, 200, 200, dtype=torch.int64),
              'instance': torch.zeros(1, 200, 200, dtype=torch.int64), 'future_egomotion': torch.zeros(1, 6),
              'hdmap': torch.zeros(1, 2, 200, 200), 'index': 100 + t}
        fr['instance'][0, 50 + 2 * t:54 + 2 * t, 60:66] = 1
        if t < rf:
            fr.update(image=torch.zeros(1, n, 3, 224, 480), intrinsics=torch.eye(3).expand(1, n, 3, 3).clone(),
                      extrinsics=torch.eye(4).expand(1, n, 4, 4).clone())
        frames.append(fr)
    data = datas.assemble_sample(frames, rf, num_instances=1, spatial_extent=(50.0, 50.0))
    assert data['image'].shape == (rf, n, 3, 224, 480) and data['intrinsics'].shape == (rf, n, 3, 3)
    assert data['extrinsics'].shape == (rf, n, 4, 4) and data['segmentation'].shape == (t_total, 1, 200, 200)
    assert data['instance'].shape == (t_total, 200, 200) and data['future_egomotion'].shape == (t_total, 6)
    assert data['hdmap'].shape == (t_total, 2, 200, 200) and data['indices'] == [100 + t for t in range(t_total)]
    assert data['centerness'].shape == (t_total, 1, 200, 200) and data['offset'].shape == (t_total, 2, 200, 200)
    assert data['flow'].shape == (t_total, 2, 200, 200) and data['target_point'].tolist() == [0.0, 0.0]
    assert set(data) >= {'image', 'intrinsics', 'extrinsics', 'segmentation', 'instance', 'centerness', 'offset', 'flow',
                         'pedestrian', 'future_egomotion', 'hdmap', 'gt_trajectory', 'indices', 'command',
                         'sample_trajectory', 'target_point'}
    # the instance moves two rows per frame: that is its displacement label (no ego motion here)
    f = data['flow']
    assert set(f[0, 0][data['instance'][0] == 1].tolist()) == {2.0} and set(f[0, 1][data['instance'][0] == 1].tolist()) == {0.0}
    assert (f[-1] == 255).all()
