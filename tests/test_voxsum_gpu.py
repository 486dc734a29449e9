"""GPU parity tests of the stand-alone voxel summing operator (stp3_amd.geometry.VoxelsSumming, through
stp3_voxels_sum_fwd / _bwd of the C ABI) against the oracle and the golden vectors generated from the
reference's own VoxelsSumming (stp3/utils/geometry.py:299-330, oracle/make_golden_voxsum.py).

Tolerances: sums vs exact float64  rtol 1e-5, atol 1e-5 (float32 sequential sum of <= 513 unit-variance terms);
            sums vs the reference at float32  atol 1e-4 (its prefix-sum differencing, recorded in the fixture);
            kept geometry rows and the backward  exact (copies).
"""
import numpy as np
import pytest
import torch

from oracle import lift_oracle as lo
from stp3_amd.geometry import VoxelsSumming
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = ('ragged', 'singles', 'onevoxel', 'onerow')


@pytest.mark.parametrize('name', CASES)
def test_against_reference_golden(name):
    g = H.load('voxsum.npz')
    x = torch.tensor(g[f'{name}_x'], device='cuda', requires_grad=True)
    geometry = torch.tensor(g[f'{name}_geometry'], device='cuda')
    ranks = torch.tensor(g[f'{name}_ranks'], device='cuda')
    out, kept = VoxelsSumming.apply(x, geometry, ranks)
    assert out.dtype == torch.float32 and tuple(out.shape) == g[f'{name}_sum64'].shape
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f'{name}_sum64'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f'{name}_sum32'], rtol=0, atol=1e-4)
    assert np.array_equal(kept.cpu().numpy(), g[f'{name}_geomkept'])
    assert not kept.requires_grad
    out.backward(torch.tensor(g[f'{name}_grad'], device='cuda'))
    assert np.array_equal(x.grad.cpu().numpy(), g[f'{name}_gradx64'])


def test_against_oracle_at_model_size_and_reproducible():
    """One (b, t) frame of the bench configuration: ~454 k in-range points in ~24 k voxels, C = 64."""
    rng = np.random.default_rng(5)
    lengths = np.minimum(rng.geometric(1.0 / 19.0, size=24000), 420)
    ranks = np.repeat(np.sort(rng.choice(40000, size=len(lengths), replace=False)), lengths).astype(np.int64)
    x = rng.standard_normal((len(ranks), 64)).astype(np.float32)
    geometry = np.stack([ranks // 200, ranks % 200, np.zeros_like(ranks)], 1)
    ref, ref_geom, seg_off = lo.voxels_summing(x, geometry, ranks)
    xt = torch.tensor(x, device='cuda', requires_grad=True)
    args = (xt, torch.tensor(geometry, device='cuda'), torch.tensor(ranks, device='cuda'))
    out, kept = VoxelsSumming.apply(*args)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    assert np.array_equal(kept.cpu().numpy(), ref_geom)
    again, _ = VoxelsSumming.apply(*args)
    assert torch.equal(out, again)                                       # fixed summation order
    grad = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.tensor(grad, device='cuda'))
    assert np.array_equal(xt.grad.cpu().numpy(), lo.voxels_summing_backward(grad, seg_off).astype(np.float32))


def test_edge_cases():
    empty = torch.zeros(0, 8, device='cuda')
    out, kept = VoxelsSumming.apply(empty, torch.zeros(0, 3, dtype=torch.long, device='cuda'),
                                    torch.zeros(0, dtype=torch.long, device='cuda'))
    assert tuple(out.shape) == (0, 8) and tuple(kept.shape) == (0, 3)
    # unsorted ranks are summed run by run, exactly like the reference's adjacent-difference mask
    x = torch.arange(12, dtype=torch.float32, device='cuda').view(6, 2)
    ranks = torch.tensor([3, 3, 1, 3, 3, 3], device='cuda')
    out, kept = VoxelsSumming.apply(x, ranks.view(6, 1).repeat(1, 3), ranks)
    assert out.cpu().tolist() == [[2.0, 4.0], [4.0, 5.0], [24.0, 27.0]]
    assert kept[:, 0].cpu().tolist() == [3, 1, 3]
    with pytest.raises(RuntimeError):
        VoxelsSumming.apply(torch.zeros(4, 2), torch.zeros(4, 3), torch.zeros(4))   # CPU tensors: no fallback
