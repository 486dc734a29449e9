"""TEST INFRASTRUCTURE (build container only) -- golden vectors of the reference's VoxelsSumming.

    python oracle/make_golden_voxsum.py

Imports the unmodified ``VoxelsSumming`` from /root/reference/stp3/utils/geometry.py (through
oracle/ref_stubs.py), runs its forward and (through autograd) backward on seeded rank-sorted
inputs, asserts that ``oracle.lift_oracle.voxels_summing`` / ``voxels_summing_backward`` agree
with it and writes tests/golden/voxsum.npz:
    <case>_x, _geometry, _ranks, _grad        inputs (float32 / int64)
    <case>_sum64, _geomkept, _gradx64         the reference evaluated in float64 (its cumsum-and-
                                              difference is exact enough there: pins the oracle)
    <case>_sum32                              the reference evaluated in float32, the precision it
                                              runs at in the model (its own rounding, recorded only)
Cases: ragged runs with long voxels, every row its own voxel, one voxel, a single row.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import lift_oracle as lo  # noqa: E402
from oracle import ref_stubs  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'voxsum.npz')


def make_case(rng, run_lengths, channels):
    ranks = np.repeat(np.sort(rng.choice(40000, size=len(run_lengths), replace=False)), run_lengths).astype(np.int64)
    m = len(ranks)
    x = rng.standard_normal((m, channels)).astype(np.float32)
    geometry = np.stack([ranks // 200, ranks % 200, np.zeros_like(ranks), rng.integers(0, 4, m)], 1).astype(np.int64)
    grad = rng.standard_normal((len(run_lengths), channels)).astype(np.float32)
    return x, geometry, ranks, grad


def main():
    ref_stubs.install()
    from stp3.utils.geometry import VoxelsSumming

    rng = np.random.default_rng(20240917)
    cases = {
        'ragged': make_case(rng, np.concatenate([rng.integers(1, 30, 100), [420, 1, 1, 257]]), 64),
        'singles': make_case(rng, np.ones(97, dtype=np.int64), 8),
        'onevoxel': make_case(rng, np.array([513]), 5),
        'onerow': make_case(rng, np.array([1]), 64),
    }
    blobs, report = {}, {}
    for name, (x, geometry, ranks, grad) in cases.items():
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        s64, gk = VoxelsSumming.apply(xt, torch.tensor(geometry), torch.tensor(ranks))
        s64.backward(torch.tensor(grad, dtype=torch.float64))
        s32, _ = VoxelsSumming.apply(torch.tensor(x), torch.tensor(geometry), torch.tensor(ranks))
        o_sum, o_geom, seg_off = lo.voxels_summing(x, geometry, ranks)
        o_gx = lo.voxels_summing_backward(grad, seg_off)
        err = float(np.abs(o_sum - s64.detach().numpy()).max())
        assert err < 1e-10, (name, err)
        assert np.array_equal(o_geom, gk.numpy()), name
        assert np.array_equal(o_gx, xt.grad.numpy()), name
        report[name] = dict(rows=int(len(ranks)), voxels=int(len(seg_off) - 1), oracle_vs_ref64=err,
                            ref32_vs_ref64=float(np.abs(s32.numpy() - s64.detach().numpy()).max()))
        blobs.update({f'{name}_x': x, f'{name}_geometry': geometry, f'{name}_ranks': ranks, f'{name}_grad': grad,
                      f'{name}_sum64': s64.detach().numpy(), f'{name}_geomkept': gk.numpy(),
                      f'{name}_gradx64': xt.grad.numpy().astype(np.float32),   # exact: copies of float32 rows
                      f'{name}_sum32': s32.numpy()})
    np.savez_compressed(OUT, **blobs)
    for k, v in report.items():
        print(k, v)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
