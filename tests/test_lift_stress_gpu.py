"""GPU: the voxel pool at BASELINE.json configs[4] geometry -- 896x1600 images (fH x fW = 112 x 200), D = 64 depth
bins, 400x400 BEV at 0.25 m (8.6 M frustum points per frame; 112-row image columns: two row slices per column in
the backward kernel, 160 000 voxels and up to ~1 M runs per frame in the plan).
B = 1, T = 2 so that ego alignment and the discounted accumulation are exercised.

Voxel ids: bit-exact against the oracle.  Pooled BEV and both gradients: against a float64 torch statement of the
same sums evaluated on the GPU from the oracle's ids (the numpy oracle would need minutes and >10 GB at this size):
rtol 1e-5 / atol 1e-5 forward, rtol 1e-4 / atol 1e-5 backward -- the tolerances of tests/test_lift_gpu.py.
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _reference(feat, logits, vox, bev_dim, discount, grad_out):
    """float64 on the GPU, camera by camera.  feat (B,T,N,C,fH,fW), logits (B,T,N,D,fH,fW), vox (B,T,N,D,fH,fW)."""
    b_, t_, n_, c_, fh, fw = feat.shape
    v_ = bev_dim[0] * bev_dim[1]
    out = torch.zeros(b_, t_, c_, v_, dtype=torch.float64, device='cuda')
    gfeat = torch.zeros_like(feat, dtype=torch.float64)
    glogit = torch.zeros_like(logits, dtype=torch.float64)
    go = grad_out.double().reshape(b_, t_, c_, v_)
    for b in range(b_):
        acc = torch.zeros(c_, v_, dtype=torch.float64, device='cuda')
        for t in range(t_):
            pool = torch.zeros(v_, c_, dtype=torch.float64, device='cuda')
            for n in range(n_):
                prob = logits[b, t, n].double().softmax(0)                             # (D,fH,fW)
                ids = vox[b, t, n].reshape(-1)
                ok = ids >= 0
                x = (prob.unsqueeze(1) * feat[b, t, n].double().unsqueeze(0))          # (D,C,fH,fW)
                x = x.permute(0, 2, 3, 1).reshape(-1, c_)
                pool.index_add_(0, ids[ok], x[ok])
            acc = acc * discount + pool.t()
            out[b, t] = acc
        g_acc = torch.zeros(c_, v_, dtype=torch.float64, device='cuda')
        for t in reversed(range(t_)):
            g_acc = g_acc * discount + go[b, t]
            gt = g_acc.t().contiguous()                                                # (V,C)
            for n in range(n_):
                prob = logits[b, t, n].double().softmax(0)
                ids = vox[b, t, n]
                g = gt[ids.clamp(min=0)] * (ids >= 0).unsqueeze(-1)                    # (D,fH,fW,C)
                f = feat[b, t, n].double()                                             # (C,fH,fW)
                dprob = torch.einsum('dhwc,chw->dhw', g, f)
                gfeat[b, t, n] = torch.einsum('dhwc,dhw->chw', g, prob)
                glogit[b, t, n] = prob * (dprob - (prob * dprob).sum(0, keepdim=True))
    return out.view(b_, t_, c_, *bev_dim), gfeat, glogit


def test_configs4_geometry_forward_backward():
    from stp3_amd import ops
    cfg = H.STRESS
    intr, extr, ego, feat, logits = H.lift_inputs(cfg, 1, 2, 6, seed=41)
    frustum, res, start, dim = H.grid_params(cfg)
    grid = ops.LiftGrid(frustum, res, start, dim, 'cuda')
    assert (grid.D, grid.fH, grid.fW, grid.X, grid.Y) == (64, 112, 200, 400, 400)
    dims = ops.make_dims(1, 2, 6, grid.D, grid.fH, grid.fW, 64, grid.X, grid.Y, grid.Z)
    mats = [m.cuda() for m in ops.lift_matrices(intr, extr, ego)]
    vox = ops.voxel_index(grid, dims, *mats, order=ops.VOX_REFERENCE).view(1, 2, 6, 64, 112, 200)
    ref_vox = H.oracle_vox(cfg, intr, extr, ego)
    assert np.array_equal(vox.cpu().numpy(), ref_vox)                                  # bit-exact ids
    assert 0.5 < float((vox >= 0).float().mean()) < 1.0
    plan = ops.LiftPlan.build(grid, intr, extr, ego, 64)
    f = feat.cuda().requires_grad_()
    lg = logits.cuda().requires_grad_()
    bev = ops.lift_splat(f, lg, plan, cfg['discount'])
    grad_out = torch.randn(bev.shape, generator=torch.Generator().manual_seed(3)).cuda()
    bev.backward(grad_out)
    exact, gf, gl = _reference(feat.cuda(), logits.cuda(), vox.long(), (400, 400), cfg['discount'], grad_out)
    torch.testing.assert_close(bev.double(), exact, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(f.grad.double(), gf, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(lg.grad.double(), gl, rtol=1e-4, atol=1e-5)
    again = ops.lift_splat(feat.cuda(), logits.cuda(), plan, cfg['discount'], True)     # channels-last memory
    assert torch.equal(again, bev.detach())                                            # fixed summation order
