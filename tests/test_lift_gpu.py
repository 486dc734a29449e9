"""GPU parity tests of the HIP lift / voxel-pool path (through the C ABI) against the oracle and
the golden fixtures generated from the reference.

Tolerances
  voxel ids                      exact (integer work)
  pooled BEV vs exact fp64 sum   rtol 1e-5, atol 1e-5   (fp32 sequential sum of <= ~400 terms)
  pooled BEV vs the reference    atol 1e-3: the reference differences a float32 prefix sum
                                 (geometry.py:305-313) and is itself only ~4e-4 accurate
                                 (tests/golden/MANIFEST.json: reference_vs_exact_fp64)
  gradients vs reference autograd / closed form   rtol 1e-4, atol 1e-5
"""
import hashlib

import numpy as np
import pytest
import torch

from oracle import lift_oracle as lo
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def _column_runs(vox):
    """vox (BT, N, D, fH, fW) -> (run-start mask, ids): a run starts where the id differs from the
    row above (same camera, depth bin and column) and is >= 0."""
    prev = np.concatenate([np.full_like(vox[:, :, :, :1], -1), vox[:, :, :, :-1]], axis=3)
    return (vox != prev) & (vox >= 0), vox


def _grid(cfg, dev='cuda'):
    from stp3_amd import ops
    frustum, res, start, dim = H.grid_params(cfg)
    return ops.LiftGrid(frustum, res, start, dim, dev)


def _ids_reference_order(cfg, intr, extr, ego):
    from stp3_amd import ops
    grid = _grid(cfg)
    b, s, n = intr.shape[:3]
    dims = ops.make_dims(b, s, n, grid.D, grid.fH, grid.fW, cfg['out_channels'], grid.X, grid.Y, grid.Z)
    mats = [m.cuda() for m in ops.lift_matrices(intr, extr, ego)]
    vox = ops.voxel_index(grid, dims, *mats, order=ops.VOX_REFERENCE)
    return vox.view(b, s, n, grid.D, grid.fH, grid.fW).cpu().numpy(), grid, dims


def _run_lift(cfg, intr, extr, ego, feat, logits, grad_out=None, channels_last=False):
    from stp3_amd import ops
    grid = _grid(cfg)
    plan = ops.LiftPlan.build(grid, intr, extr, ego, cfg['out_channels'])
    f = feat.cuda().requires_grad_(grad_out is not None)
    l = logits.cuda().requires_grad_(grad_out is not None)
    bev = ops.lift_splat(f, l, plan, cfg['discount'], channels_last)
    assert (bev.permute(0, 1, 3, 4, 2) if channels_last else bev).is_contiguous()
    grads = None
    if grad_out is not None:
        bev.backward(grad_out.cuda())
        grads = (f.grad.cpu(), l.grad.cpu())
    return bev.detach().cpu(), plan, grads


def test_small_case_against_reference_golden():
    g = H.load('lift_small.npz')
    intr, extr, ego = (torch.from_numpy(g[k]) for k in ('intrinsics', 'extrinsics', 'future_egomotion'))
    feat, logits = torch.from_numpy(g['feat']), torch.from_numpy(g['depth_logits'])
    vox, grid, dims = _ids_reference_order(H.SMALL, intr, extr, ego)
    assert np.array_equal(vox, g['ref_vox'])
    bev, plan, grads = _run_lift(H.SMALL, intr, extr, ego, feat, logits, torch.from_numpy(g['grad_out']))
    # the plan's column-major ids are the same ids, permuted
    assert np.array_equal(plan.voxel_ids().cpu().numpy(), g['ref_vox'])
    # plan structure: runs (maximal stretches of equal voxel id along an image column) counted per voxel (exclusive
    # scan = histogram of run starts); the work groups of the forward kernel -- ascending, cover [0, V), <= 16 voxels
    # each inside one 16-voxel block, and (unless a single voxel) inside one work bucket of 32 runs over the sample's
    # frames; every (group, frame) range of the descriptor list holds exactly the runs of the group's voxels, longest
    # first, ties by (camera, column, depth bin, first row)
    starts, ids = _column_runs(g['ref_vox'].reshape(6, 2, grid.D, grid.fH, grid.fW))
    off = plan.offsets().cpu().numpy()
    desc = plan.descriptors().cpu().numpy()
    runs = []
    for bt in range(6):
        hist = np.bincount(ids[bt][starts[bt]], minlength=dims.V)
        assert np.array_equal(np.diff(off[bt]), hist)
        assert off[bt][-1] == starts[bt].sum()
        per_voxel = {}
        nn, dd, hh, ww = np.nonzero(starts[bt])
        for n_, d_, h_, w_ in zip(nn, dd, hh, ww):
            v = int(ids[bt][n_, d_, h_, w_])
            ln = 1
            while h_ + ln < grid.fH and ids[bt][n_, d_, h_ + ln, w_] == v and not starts[bt][n_, d_, h_ + ln, w_]:
                ln += 1
            per_voxel.setdefault(v, []).append((-ln, ((n_ * grid.fW + w_) << 20) | (d_ << 14) | (h_ << 7) | (ln - 1), v))
        runs.append(per_voxel)
    work = np.diff(off.reshape(2, 3, -1), axis=2).sum(axis=1)            # runs per voxel, all frames of a sample
    for b, gl in enumerate(plan.groups()):
        gl = gl.cpu().numpy()
        assert gl[0] == 0 and gl[-1] == dims.V and (np.diff(gl) > 0).all() and np.diff(gl).max() <= 16
        assert ((gl[:-1] // 16) == ((gl[1:] - 1) // 16)).all()
        before = np.concatenate([[0], np.cumsum(work[b])])
        for a, e in zip(gl[:-1], gl[1:]):
            assert before[e - 1] // 32 == before[a] // 32                   # all voxels of a group start in one bucket
            for t in range(3):
                bt = b * 3 + t
                want = sorted(r for v in range(a, e) for r in runs[bt].get(v, []))
                rows = desc[bt][off[bt][a]:off[bt][e]]
                assert [(int(r[0]), int(r[1])) for r in rows] == [(k, v) for _, k, v in want]
                for k, v, frow, prow in rows:                              # the two precomputed addresses
                    col, d_, h_ = int(k) >> 20, (int(k) >> 14) & 63, (int(k) >> 7) & 127
                    n_, w_ = divmod(col, grid.fW)
                    assert frow == (n_ * grid.fH + h_) * grid.fW + w_ and prow == (col * grid.D + d_) * grid.fH + h_
    assert int(plan.counts.abs().max()) == 0                              # the scratch is left clean for the next build
    exact = lo.pool_exact(feat, logits, g['ref_vox'], (32, 32), 0.5)
    torch.testing.assert_close(bev.double(), exact, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bev, torch.from_numpy(g['ref_bev']), rtol=0, atol=1e-3)
    torch.testing.assert_close(grads[0], torch.from_numpy(g['ref_grad_feat']), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(grads[1], torch.from_numpy(g['ref_grad_logits']), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name,axis', [('generic', False), ('axis_aligned', True)])
def test_full_size_ids_bit_exact(name, axis):
    g = H.load('lift_full.npz')
    intr, extr, ego, feat, logits = H.lift_inputs(H.FULL, 1, 3, 6, seed=5, axis_aligned=axis)
    vox, _, _ = _ids_reference_order(H.FULL, intr, extr, ego)
    assert np.array_equal(vox.reshape(-1)[::97], g[f'{name}_vox_sample'])
    assert np.array_equal(_sha(vox), g[f'{name}_vox_sha256'])
    assert np.array_equal(vox, H.oracle_vox(H.FULL, intr, extr, ego))


@pytest.mark.parametrize('name,axis', [('generic', False), ('axis_aligned', True)])
def test_full_size_pool_forward_backward(name, axis):
    g = H.load('lift_full.npz')
    intr, extr, ego, feat, logits = H.lift_inputs(H.FULL, 1, 3, 6, seed=5, axis_aligned=axis)
    gen = torch.Generator().manual_seed(17)
    grad_out = torch.randn(1, 3, 64, 200, 200, generator=gen)
    bev, plan, grads = _run_lift(H.FULL, intr, extr, ego, feat, logits, grad_out)
    flat = bev.reshape(-1)
    # (1) vs the exact per-voxel sum computed from the reference-validated ids
    torch.testing.assert_close(flat[::257].double(), torch.from_numpy(g[f'{name}_bev_exact_sample']),
                               rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bev.double().sum(dim=(-1, -2))[0], torch.from_numpy(g[f'{name}_bev_sum_tc']),
                               rtol=1e-5, atol=1e-3)
    # (2) vs what the reference itself produced (lossy prefix-sum trick)
    torch.testing.assert_close(flat[::257], torch.from_numpy(g[f'{name}_bev_sample']), rtol=0, atol=1e-3)
    # (3) everything, against the oracle run here
    vox = H.oracle_vox(H.FULL, intr, extr, ego)
    exact = lo.pool_exact(feat, logits, vox, (200, 200), 0.5)
    torch.testing.assert_close(bev.double(), exact, rtol=1e-5, atol=1e-5)
    gf, gl = lo.pool_backward_exact(grad_out, feat, logits, vox, 0.5)
    torch.testing.assert_close(grads[0].double(), gf, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(grads[1].double(), gl, rtol=1e-4, atol=1e-5)


def test_forward_is_bit_reproducible_and_layout_independent():
    intr, extr, ego, feat, logits = H.lift_inputs(H.FULL, 2, 3, 6, seed=23)
    gen = torch.Generator().manual_seed(29)
    grad_out = torch.randn(2, 3, 64, 200, 200, generator=gen)
    a, _, ga = _run_lift(H.FULL, intr, extr, ego, feat, logits, grad_out)
    b, _, gb = _run_lift(H.FULL, intr, extr, ego, feat, logits, grad_out)
    assert torch.equal(a, b) and torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1])
    # channels-last BEV (what the model uses): the same numbers, bit for bit, in [B,T,X,Y,C] memory, and the same
    # gradients from a gradient that arrives in that layout
    c, _, gc = _run_lift(H.FULL, intr, extr, ego, feat, logits, grad_out, channels_last=True)
    assert c.shape == a.shape and torch.equal(a, c)
    assert torch.equal(ga[0], gc[0]) and torch.equal(ga[1], gc[1])


def test_batch4_properties_at_bench_size():
    """BASELINE.json configs[1]/[2] shape (B=4, T=3): size-independent properties."""
    intr, extr, ego, feat, logits = H.lift_inputs(H.FULL, 4, 3, 6, seed=31)
    bev, plan, _ = _run_lift(H.FULL, intr, extr, ego, feat, logits)
    # samples are independent: sample 2 alone gives the same bits
    one, _, _ = _run_lift(H.FULL, intr[2:3], extr[2:3], ego[2:3], feat[2:3], logits[2:3])
    assert torch.equal(bev[2:3], one)
    # mass conservation: sum over voxels of frame-0 output = sum over in-range points of prob*feat
    vox_pm = plan.voxel_ids().permute(0, 1, 2, 4, 5, 3).reshape(4, 3, -1, 48).cpu()      # (B,T,NPIX,D)
    prob = logits.permute(0, 1, 2, 4, 5, 3).reshape(4, 3, -1, 48).softmax(-1).double()
    w = (prob * (vox_pm >= 0)).sum(-1)                                  # (B,T,NPIX)
    f = feat.permute(0, 1, 2, 4, 5, 3).reshape(4, 3, -1, 64).double()
    mass = torch.einsum('btp,btpc->btc', w, f)
    torch.testing.assert_close(bev[:, 0].double().sum(dim=(-1, -2)), mass[:, 0], rtol=1e-6, atol=1e-4)
    # linearity in feat and the discount recurrence out[t] = 0.5*out[t-1] + Pool_t
    bev2, _, _ = _run_lift(H.FULL, intr, extr, ego, feat * 2.0, logits)
    assert torch.equal(bev2, bev * 2.0)
    pool2 = bev[:, 2] - 0.5 * bev[:, 1]
    torch.testing.assert_close(pool2.double().sum(dim=(-1, -2)), mass[:, 2], rtol=1e-5, atol=1e-3)


def test_edge_cases():
    from stp3_amd import ops
    # T = 1 (BASELINE.json configs[0]) and a single camera
    intr, extr, ego, feat, logits = H.lift_inputs(H.SMALL, 1, 1, 1, seed=3)
    bev, plan, _ = _run_lift(H.SMALL, intr, extr, ego, feat, logits)
    vox = H.oracle_vox(H.SMALL, intr, extr, ego)
    torch.testing.assert_close(bev.double(), lo.pool_exact(feat, logits, vox, (32, 32), 0.5), rtol=1e-5, atol=1e-5)
    # every point outside the grid -> all zeros, empty lists
    far = extr.clone()
    far[..., :3, 3] += 1000.0
    bev, plan, _ = _run_lift(H.SMALL, intr, far, ego, feat, logits)
    assert (plan.vox_cm == -1).all() and (bev == 0).all()
    # NaN pose: the points are dropped exactly like the reference's `.long()` + mask does
    bad = extr.clone()
    bad[0, 0, 0, 0, 3] = float('nan')
    _, plan, _ = _run_lift(H.SMALL, intr, bad, ego, feat, logits)
    assert (plan.vox_cm == -1).all()
    # unsupported shapes are rejected, not mis-computed
    grid = _grid(H.SMALL)
    with pytest.raises(Exception):
        ops.lift_splat(feat, logits, plan, 0.5)          # CPU tensors: no fallback


def test_many_points_in_one_voxel():
    """Collision stress: a coarse grid makes every voxel list long (thousands of runs per voxel: the per-voxel
    ordering pass of the plan and the run loop of the forward kernel at their worst)."""
    cfg = dict(H.FULL, x_bound=(-50.0, 50.0, 12.5), y_bound=(-50.0, 50.0, 12.5))
    intr, extr, ego, feat, logits = H.lift_inputs(cfg, 1, 2, 6, seed=41)
    bev, plan, _ = _run_lift(cfg, intr, extr, ego, feat, logits)
    again, _, _ = _run_lift(cfg, intr, extr, ego, feat, logits)
    vox = H.oracle_vox(cfg, intr, extr, ego)
    exact = lo.pool_exact(feat, logits, vox, (8, 8), 0.5)
    torch.testing.assert_close(bev.double(), exact, rtol=2e-4, atol=1e-3)   # up to ~50k terms per sum
    counts = np.bincount(vox[0, 0][vox[0, 0] >= 0])
    assert counts.max() > 4096
    assert torch.equal(bev, again)                                         # canonical order at any list length
