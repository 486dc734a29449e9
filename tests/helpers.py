"""Shared test fixtures: seeded inputs (identical to oracle/make_golden.py) and golden loaders."""
import os

import numpy as np
import torch

from oracle import lift_oracle as lo
from stp3_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

SMALL = dict(final_dim=(32, 48), x_bound=(-8.0, 8.0, 0.5), y_bound=(-8.0, 8.0, 0.5), z_bound=(-10.0, 10.0, 20.0),
             d_bound=(2.0, 10.0, 1.0), downsample=8, out_channels=8, discount=0.5)
FULL = dict(final_dim=(224, 480), x_bound=(-50.0, 50.0, 0.5), y_bound=(-50.0, 50.0, 0.5),
            z_bound=(-10.0, 10.0, 20.0), d_bound=(2.0, 50.0, 1.0), downsample=8, out_channels=64, discount=0.5)
# BASELINE.json configs[4]: 896x1600 (900 is not divisible by 16, SURVEY.md hard part 12), D=64, 400x400 BEV
STRESS = dict(final_dim=(896, 1600), x_bound=(-50.0, 50.0, 0.25), y_bound=(-50.0, 50.0, 0.25),
              z_bound=(-10.0, 10.0, 20.0), d_bound=(2.0, 66.0, 1.0), downsample=8, out_channels=64, discount=0.5)


def lift_inputs(cfg, batch, seq, n_cams, seed, axis_aligned=False):
    h, w = cfg['final_dim']
    fh, fw = h // cfg['downsample'], w // cfg['downsample']
    d = int((cfg['d_bound'][1] - cfg['d_bound'][0]) / cfg['d_bound'][2])
    intr, extr, ego = synthetic.make_rig(batch, seq, n_cams, cfg['final_dim'], seed=seed, axis_aligned=axis_aligned)
    g = torch.Generator().manual_seed(seed + 1)
    feat = torch.relu(torch.randn(batch, seq, n_cams, cfg['out_channels'], fh, fw, generator=g))
    logits = torch.randn(batch, seq, n_cams, d, fh, fw, generator=g) * 2.0
    return intr, extr, ego, feat, logits


def grid_params(cfg):
    frustum = lo.create_frustum(cfg['final_dim'], cfg['downsample'], list(cfg['d_bound']))
    res, start, dim = lo.bev_parameters(cfg['x_bound'], cfg['y_bound'], cfg['z_bound'])
    return frustum, res, start, dim


def oracle_vox(cfg, intr, extr, ego):
    frustum, _, _, _ = grid_params(cfg)
    return lo.lift_voxel_ids(frustum, intr, extr, ego, cfg['x_bound'], cfg['y_bound'], cfg['z_bound'])


def load(name):
    return np.load(os.path.join(GOLDEN, name))
