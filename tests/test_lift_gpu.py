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
    # plan structure: row masks, run slots, per-voxel slot lists (ascending), scratch left clean
    assert H.check_plan_structure(plan, g['ref_vox']) > 0
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


def test_bf16_bev_output_is_the_float32_result_rounded_once():
    """STP3_BEV_CHANNELS_LAST_BF16 (what the bf16 training step uses): bit-equal to rounding the float32 channels-last
    BEV to bf16, and the backward of a bf16 gradient gives the same bits as through the float32 output."""
    from stp3_amd import ops
    intr, extr, ego, feat, logits = H.lift_inputs(H.FULL, 2, 3, 6, seed=23)
    frustum, res, start, dim = H.grid_params(H.FULL)
    grid = ops.LiftGrid(frustum, res, start, dim, 'cuda')
    plan = ops.LiftPlan.build(grid, intr, extr, ego, 64)
    go = torch.randn(2, 3, 200, 200, 64, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).cuda().permute(0, 1, 4, 2, 3)
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        f, l = feat.cuda().requires_grad_(), logits.cuda().requires_grad_()
        bev = ops.lift_splat(f, l, plan, 0.5, True, dtype)
        assert bev.dtype == dtype and bev.permute(0, 1, 3, 4, 2).is_contiguous()
        bev.backward(go if dtype == torch.bfloat16 else go)
        out.append((bev.detach(), f.grad, l.grad))
    (b32, gf32, gl32), (b16, gf16, gl16) = out
    assert torch.equal(b16, b32.to(torch.bfloat16))
    assert torch.equal(gf16, gf32) and torch.equal(gl16, gl32)


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
