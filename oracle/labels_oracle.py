"""TEST INFRASTRUCTURE -- CPU oracle of the BEV label path.  Only tests/ may import this.

``fill_poly``: cv2.fillPoly (third-party, NOT installed here: PARITY UNPINNED) restated pixel by pixel from OpenCV's
published algorithm (modules/imgproc/src/drawing.cpp): a pixel is painted when it lies on the 8-connected Bresenham
outline (``LineIterator`` walked left to right: error term M - 2 m, a diagonal step whenever it is negative) or, on a
scanline y0 <= y < y1 of the non-horizontal edges, between a pair of active edges sorted by x -- ceil(left) .. floor(right)
with the edge positions in 16.16 fixed point and the slope truncated to that precision (``FillEdgeCollection``).  Brute
force over the whole image in Python integers: slow and obviously right; the product has the same algorithm twice more
(edge walking in stp3_amd/datas.py, closed form per pixel in csrc/stp3_labels.hip).

``instance_labels``: stp3/utils/instance.py:12-77 is importable here -- the fixture tests/golden/labels.npz comes from the
reference's own function (oracle/make_golden_labels.py); no restatement needed."""
import numpy as np


def bresenham(x0, y0, x1, y1):
    """Pixels of OpenCV's 8-connected LineIterator from (x0, y0) to (x1, y1), drawn left to right."""
    if x0 > x1:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, y1 - y0
    sy = -1 if dy < 0 else 1
    ady = abs(dy)
    steep = ady > dx
    big, small = (ady, dx) if steep else (dx, ady)
    err = big - 2 * small
    x, y, out = x0, y0, []
    for _ in range(big + 1):
        out.append((x, y))
        diag = err < 0
        err += -2 * small + (2 * big if diag else 0)
        if steep:
            y += sy
            x += 1 if diag else 0
        else:
            x += 1
            y += sy if diag else 0
    return out


def _trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def fill_poly(img, poly, value):
    """In place: cv2.fillPoly(img, [poly], value) for integer (column, row) vertices; img (H, W)."""
    h, w = img.shape
    poly = [(int(px), int(py)) for px, py in poly]
    outline = set()
    edges = []
    for v in range(len(poly)):
        (ax, ay), (bx, by) = poly[v - 1], poly[v]
        outline.update(bresenham(ax, ay, bx, by))
        if ay != by:
            y0, y1, x0 = (ay, by, ax) if ay < by else (by, ay, bx)
            edges.append((y0, y1, x0 << 16, _trunc_div((bx - ax) << 16, by - ay)))
    for y in range(h):
        xs = sorted(x0 + slope * (y - y0) for y0, y1, x0, slope in edges if y0 <= y < y1)
        for x in range(w):
            paint = (x, y) in outline
            for left, right in zip(xs[0::2], xs[1::2]):
                paint = paint or ((left + 65535) >> 16) <= x <= (right >> 16)
            if paint:
                img[y, x] = value
    return img


def fill_polygons(polys, values, map_index, n_maps, hw):
    maps = np.zeros((n_maps,) + tuple(hw), dtype=np.float32)
    for poly, value, mi in zip(polys, values, map_index):
        fill_poly(maps[mi], poly, value)
    return maps
