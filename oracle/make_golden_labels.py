"""TEST INFRASTRUCTURE (build container only) -- fixtures of the BEV label path.

    python oracle/make_golden_labels.py        -> tests/golden/labels.npz

  * ``instance/...``: the reference's own ``convert_instance_mask_to_center_and_offset_label`` (stp3/utils/instance.py:
    12-77, imported unmodified through oracle/ref_stubs.py) on a synthetic sequence: T = 5 frames of a 200 x 200 BEV with 14
    box-shaped instances (painted by oracle.labels_oracle.fill_poly) that move, turn, appear and vanish, under an ego
    motion; inputs and the three label tensors are stored (centerness as float32, offsets / displacements exactly).
  * ``poly/...``: 40 annotation-like boxes (bottom corners in the ego frame -> ``box_polygons``) and what three
    rasterisers paint for them: the oracle's restatement of cv2.fillPoly (third-party, NOT installed: parity unpinned) and
    Pillow's ``ImageDraw.polygon`` with fill + outline (third-party, installed: an independent cross-check; it is NOT
    cv2, so it may differ on outline pixels -- the fixture records how often).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

from oracle import labels_oracle as lo  # noqa: E402
from oracle import ref_stubs  # noqa: E402
from stp3_amd import datas  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
BEV_START, BEV_RES, BEV_DIM = (-49.75, -49.75, 0.0), (0.5, 0.5, 20.0), (200, 200, 1)


def boxes(rng, n, spread=42.0):
    centres = rng.uniform(-spread, spread, (n, 2))
    length, width = rng.uniform(1.5, 12.0, n), rng.uniform(0.6, 3.0, n)
    yaw = rng.uniform(0, np.pi, n)
    out = []
    for c, l, w, a in zip(centres, length, width, yaw):
        rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        out.append(np.array([[l, w], [l, -w], [-l, -w], [-l, w]]) / 2 @ rot.T + c)
    return np.stack(out)


def main():
    ref_stubs.install()
    from stp3.utils.instance import convert_instance_mask_to_center_and_offset_label
    rng = np.random.default_rng(11)
    out = {}
    # ---- instance sequence
    t_, k = 5, 14
    base = boxes(rng, k, spread=38.0)
    vel = rng.uniform(-3.0, 3.0, (k, 2))
    instance = np.zeros((t_, 200, 200), dtype=np.float32)
    for t in range(t_):
        for i in range(k):
            if (i == 3 and t >= 3) or (i == 7 and t == 2) or (i == 11 and t == 0):       # vanish / skip a frame / appear late
                continue
            poly = datas.box_polygons(base[i] + vel[i] * t, BEV_START, BEV_RES)
            lo.fill_poly(instance[t], poly, float(i + 1))
    ego = torch.tensor([[1.8, 0.1, 0.0, 0.0, 0.0, 0.03], [2.1, -0.2, 0.0, 0.0, 0.0, -0.02], [1.5, 0.0, 0.0, 0.0, 0.0, 0.05],
                        [2.4, 0.3, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]], dtype=torch.float32)
    inst = torch.from_numpy(instance).long()
    center, offset, flow = convert_instance_mask_to_center_and_offset_label(
        inst, ego, num_instances=k, ignore_index=255, subtract_egomotion=True, spatial_extent=(50.0, 50.0))
    out['instance/ids'] = inst.numpy().astype(np.int16)
    out['instance/future_egomotion'] = ego.numpy()
    out['instance/num_instances'] = np.array([k])
    out['instance/center'] = center.numpy().astype(np.float32)
    out['instance/offset'] = offset.numpy().astype(np.float32)
    out['instance/flow'] = flow.numpy().astype(np.float32)
    # ---- polygons: oracle restatement of cv2.fillPoly and Pillow's polygon
    from PIL import Image, ImageDraw
    corners = boxes(rng, 40, spread=52.0)                                # some of them cross the border of the grid
    polys = datas.box_polygons(corners, BEV_START, BEV_RES)
    oracle_maps = np.zeros((40, 200, 200), dtype=np.uint8)
    pillow_maps = np.zeros((40, 200, 200), dtype=np.uint8)
    for i, poly in enumerate(polys):
        img = np.zeros((200, 200), dtype=np.float32)
        oracle_maps[i] = lo.fill_poly(img, poly, 1.0).astype(np.uint8)
        pim = Image.new('L', (200, 200), 0)
        ImageDraw.Draw(pim).polygon([tuple(int(v) for v in p) for p in poly], fill=1, outline=1)
        pillow_maps[i] = np.asarray(pim)
    out['poly/corners'] = corners
    out['poly/vertices'] = polys.astype(np.int32)
    out['poly/oracle'] = np.packbits(oracle_maps.reshape(40, -1), axis=1)
    out['poly/pillow'] = np.packbits(pillow_maps.reshape(40, -1), axis=1)
    diff = (oracle_maps != pillow_maps)
    np.savez_compressed(os.path.join(GOLDEN, 'labels.npz'), **out)
    man_path = os.path.join(GOLDEN, 'MANIFEST.json')
    man = json.load(open(man_path))
    man['labels'] = {'file': 'labels.npz', 'generator': 'oracle/make_golden_labels.py',
                     'instance': 'reference convert_instance_mask_to_center_and_offset_label, T=5, 14 instances, float32 CPU',
                     'polygons': {'count': 40, 'painted_pixels_oracle': int(oracle_maps.sum()),
                                  'painted_pixels_pillow': int(pillow_maps.sum()), 'pixels_that_differ': int(diff.sum()),
                                  'polygons_that_differ': int(diff.reshape(40, -1).any(1).sum()),
                                  'only_oracle': int((oracle_maps > pillow_maps).sum()),
                                  'only_pillow': int((pillow_maps > oracle_maps).sum())},
                     'unpinned_third_party': ['opencv-python cv2.fillPoly (restated; cross-checked against Pillow)']}
    json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)
    print(json.dumps(man['labels'], indent=1))


if __name__ == '__main__':
    main()
