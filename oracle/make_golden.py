"""TEST INFRASTRUCTURE (build container only) -- generate tests/golden/* from the REFERENCE itself.

    python oracle/make_golden.py

Imports the unmodified reference from /root/reference (through oracle/ref_stubs.py), runs its own
``STP3.calculate_birds_eye_view_features`` (stp3/models/stp3.py:303-318: get_geometry ->
encoder_forward tail -> projection_to_birds_eye_view) with a stand-in encoder that returns fixed
feature / depth-logit tensors, and
  1. asserts that oracle/lift_oracle.py reproduces the reference (voxel ids and geometry bitwise,
     pooled values bitwise for the literal restatement, gradients to 1e-6), and
  2. writes the fixtures the tests replay where /root/reference does not exist (the GPU box):
       tests/golden/lift_small.npz     complete tensors of a small configuration
       tests/golden/lift_full.npz      224x480 / 6 cameras / T=3 (generic + axis-aligned rigs):
                                       sha256 + strided samples of the ids, strided samples of
                                       the pooled output, per-(t,c) sums
       tests/golden/MANIFEST.json      what was checked, with the measured deviations
Inputs come from stp3_amd.synthetic (seeded CPU generators), so tests regenerate them bit-for-bit.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

from oracle import lift_oracle as lo  # noqa: E402
from oracle import ref_stubs  # noqa: E402
from stp3_amd import synthetic  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

SMALL = dict(final_dim=(32, 48), x_bound=(-8.0, 8.0, 0.5), y_bound=(-8.0, 8.0, 0.5), z_bound=(-10.0, 10.0, 20.0),
             d_bound=(2.0, 10.0, 1.0), downsample=8, out_channels=8, discount=0.5)
FULL = dict(final_dim=(224, 480), x_bound=(-50.0, 50.0, 0.5), y_bound=(-50.0, 50.0, 0.5),
            z_bound=(-10.0, 10.0, 20.0), d_bound=(2.0, 50.0, 1.0), downsample=8, out_channels=64, discount=0.5)


def lift_inputs(cfg, batch, seq, n_cams, seed, axis_aligned=False):
    """Seeded inputs shared by this script and tests/ (see tests/helpers.py)."""
    h, w = cfg['final_dim']
    fh, fw = h // cfg['downsample'], w // cfg['downsample']
    d = int((cfg['d_bound'][1] - cfg['d_bound'][0]) / cfg['d_bound'][2])
    intr, extr, ego = synthetic.make_rig(batch, seq, n_cams, cfg['final_dim'], seed=seed, axis_aligned=axis_aligned)
    g = torch.Generator().manual_seed(seed + 1)
    feat = torch.relu(torch.randn(batch, seq, n_cams, cfg['out_channels'], fh, fw, generator=g))
    logits = torch.randn(batch, seq, n_cams, d, fh, fw, generator=g) * 2.0
    return intr, extr, ego, feat, logits


class _FixedEncoder(torch.nn.Module):
    """Stands in for stp3.models.encoder.Encoder: returns the packed (B*S*N, C|D, fH, fW) tensors."""

    def __init__(self, feat, logits):
        super().__init__()
        self.feat, self.logits = feat, logits

    def forward(self, x):
        return self.feat.reshape(-1, *self.feat.shape[3:]), self.logits.reshape(-1, *self.logits.shape[3:])


def run_reference(cfg, intr, extr, ego, feat, logits, grad_out=None):
    m = ref_stubs.make_reference_lifter(**cfg)
    feat = feat.clone().requires_grad_(grad_out is not None)
    logits = logits.clone().requires_grad_(grad_out is not None)
    m.encoder = _FixedEncoder(feat, logits)
    b, s, n = intr.shape[:3]
    images = torch.zeros(b, s, n, 3, *cfg['final_dim'])
    # geometry as the reference computes it (for the id comparison); recomputed inside the call below
    geo = m.get_geometry(intr.view(b * s, n, 3, 3), extr.view(b * s, n, 4, 4)).view(b, s, n, *m.frustum.shape)
    bev, depth, _ = m.calculate_birds_eye_view_features(images, intr, extr, ego)
    grads = None
    if grad_out is not None:
        bev.backward(grad_out)
        grads = (feat.grad.clone(), logits.grad.clone())
    # reference voxel ids: replay stp3.py:270-277 + 287-289 + 239-255 with the reference's own ops
    from stp3.utils.geometry import pose_vec2mat
    pm = pose_vec2mat(ego)
    rot, tr = pm[..., :3, :3], pm[..., :3, 3]
    geo = geo.clone()
    for bi in range(b):
        fg = geo[bi]
        for t in range(s):
            if t != s - 1:
                tmp = rot[bi, t].view(1, 1, 1, 1, 1, 3, 3).matmul(fg[:t + 1].unsqueeze(-1)).squeeze(-1)
                tmp += tr[bi, t].view(1, 1, 1, 1, 1, 3)
                fg[:t + 1] = tmp
    gi = ((geo - (m.bev_start_position - m.bev_resolution / 2.0)) / m.bev_resolution).long()
    dim = m.bev_dimension
    keep = ((gi[..., 0] >= 0) & (gi[..., 0] < dim[0]) & (gi[..., 1] >= 0) & (gi[..., 1] < dim[1])
            & (gi[..., 2] >= 0) & (gi[..., 2] < dim[2]))
    rank = gi[..., 0] * (dim[1] * dim[2]) + gi[..., 1] * dim[2] + gi[..., 2]
    vox = torch.where(keep, rank, torch.full_like(rank, -1)).to(torch.int32).numpy()
    return bev.detach(), depth.detach(), vox, grads, geo.numpy()


def run_oracle(cfg, intr, extr, ego, feat, logits):
    fr = lo.create_frustum(cfg['final_dim'], cfg['downsample'], list(cfg['d_bound']))
    vox = lo.lift_voxel_ids(fr, intr, extr, ego, cfg['x_bound'], cfg['y_bound'], cfg['z_bound'])
    _, _, dim = lo.bev_parameters(cfg['x_bound'], cfg['y_bound'], cfg['z_bound'])
    bev_ref_style = lo.pool_reference_style(feat, logits, vox, dim.tolist(), cfg['discount'])
    bev_exact = lo.pool_exact(feat, logits, vox, dim.tolist(), cfg['discount'])
    return vox, bev_ref_style, bev_exact


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {'torch': torch.__version__, 'numpy': np.__version__, 'threads': torch.get_num_threads(),
                'reference': 'OpenDriveLab/ST-P3 @ /root/reference (stp3/models/stp3.py, stp3/utils/geometry.py)',
                'cases': {}}

    # ------------------------------------------------------------------ small, complete tensors
    b, s, n = 2, 3, 2
    intr, extr, ego, feat, logits = lift_inputs(SMALL, b, s, n, seed=11)
    g = torch.Generator().manual_seed(99)
    grad_out = torch.randn(b, s, SMALL['out_channels'], 32, 32, generator=g)
    bev, depth, vox, grads, _ = run_reference(SMALL, intr, extr, ego, feat, logits, grad_out)
    o_vox, o_ref_style, o_exact = run_oracle(SMALL, intr, extr, ego, feat, logits)
    assert np.array_equal(vox, o_vox), 'oracle voxel ids differ from the reference (small)'
    assert torch.equal(o_ref_style, bev), 'literal pooling restatement differs from the reference (small)'
    og_feat, og_logit = lo.pool_backward_exact(grad_out, feat, logits, o_vox, SMALL['discount'])
    gf_err = (og_feat - grads[0].double()).abs().max().item()
    gl_err = (og_logit - grads[1].double()).abs().max().item()
    assert gf_err < 1e-5 and gl_err < 1e-5, (gf_err, gl_err)
    np.savez_compressed(os.path.join(GOLDEN, 'lift_small.npz'),
                        intrinsics=intr.numpy(), extrinsics=extr.numpy(), future_egomotion=ego.numpy(),
                        feat=feat.numpy(), depth_logits=logits.numpy(), grad_out=grad_out.numpy(),
                        ref_vox=vox, ref_bev=bev.numpy(), ref_grad_feat=grads[0].numpy(),
                        ref_grad_logits=grads[1].numpy())
    manifest['cases']['lift_small'] = {
        'config': {k: list(v) if isinstance(v, tuple) else v for k, v in SMALL.items()},
        'shape': {'B': b, 'T': s, 'N': n}, 'seed': 11,
        'oracle_vox_equal_reference': True, 'oracle_literal_pool_bitwise_equal_reference': True,
        'oracle_exact_vs_reference_max_abs': (o_exact - bev.double()).abs().max().item(),
        'oracle_closed_form_grad_vs_reference_autograd_max_abs': {'feat': gf_err, 'logits': gl_err},
        'in_range_fraction': float((vox >= 0).mean()),
    }

    # ------------------------------------------------------------------ full size, digests + samples
    full = {}
    for name, axis in (('generic', False), ('axis_aligned', True)):
        b, s, n = 1, 3, 6
        intr, extr, ego, feat, logits = lift_inputs(FULL, b, s, n, seed=5, axis_aligned=axis)
        bev, depth, vox, _, geo = run_reference(FULL, intr, extr, ego, feat, logits)
        o_vox, o_ref_style, o_exact = run_oracle(FULL, intr, extr, ego, feat, logits)
        assert np.array_equal(vox, o_vox), f'oracle voxel ids differ from the reference ({name})'
        bitwise = bool(torch.equal(o_ref_style, bev))
        dev = (bev.double() - o_exact).abs()
        rel = dev / o_exact.abs().clamp(min=1e-12)
        nz = o_exact != 0
        flat = bev.numpy().reshape(-1)
        full[f'{name}_vox_sha256'] = np.frombuffer(bytes.fromhex(sha(vox)), dtype=np.uint8)
        full[f'{name}_vox_sample'] = vox.reshape(-1)[::97].copy()
        full[f'{name}_vox_hist_sha256'] = np.frombuffer(bytes.fromhex(sha(
            np.stack([np.bincount(vox[0, t][vox[0, t] >= 0], minlength=40000) for t in range(s)]).astype(np.int32))),
            dtype=np.uint8)
        full[f'{name}_bev_sample'] = flat[::257].copy()
        full[f'{name}_bev_exact_sample'] = o_exact.numpy().reshape(-1)[::257].astype(np.float64)
        full[f'{name}_bev_sum_tc'] = o_exact.sum(dim=(-1, -2)).numpy()[0]
        manifest['cases'][f'lift_full_{name}'] = {
            'config': {k: list(v) if isinstance(v, tuple) else v for k, v in FULL.items()},
            'shape': {'B': b, 'T': s, 'N': n}, 'seed': 5, 'axis_aligned': axis,
            'oracle_vox_equal_reference': True,
            'oracle_literal_pool_bitwise_equal_reference': bitwise,
            'oracle_literal_pool_vs_reference_max_abs': (o_ref_style - bev).abs().max().item(),
            'reference_vs_exact_fp64': {'max_abs': dev.max().item(),
                                        'frac_rel_gt_1e-3': float((rel[nz] > 1e-3).float().mean())},
            'in_range_fraction': float((vox >= 0).mean()),
            'occupied_voxels_per_frame': [int((np.bincount(vox[0, t][vox[0, t] >= 0], minlength=40000) > 0).sum())
                                          for t in range(s)],
            'max_points_per_voxel': int(max(np.bincount(vox[0, t][vox[0, t] >= 0]).max() for t in range(s))),
        }
    np.savez_compressed(os.path.join(GOLDEN, 'lift_full.npz'), **full)

    with open(os.path.join(GOLDEN, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
