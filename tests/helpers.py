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


# ----------------------------------------------------------------------------------------------
# deterministic weights: identical on every box, independent of construction order / RNG
# ----------------------------------------------------------------------------------------------
def fill_deterministic(module, seed=0):
    """Overwrite every floating tensor of ``module.state_dict()`` with values derived from the
    tensor's *name* by exact integer arithmetic (an LCG), so the reference modules (in the build
    container) and the modules under test (on the GPU box) get bit-identical weights."""
    import math
    import zlib
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if not t.is_floating_point():
                continue
            if name.split('.')[-1] in ('frustum', 'bev_resolution', 'bev_start_position', 'bev_dimension'):
                continue                                         # geometry constants, not weights
            n = t.numel()
            h = (zlib.crc32(name.encode()) + 7919 * seed) % 2147483647
            idx = torch.arange(n, dtype=torch.int64)
            u = ((idx * 1103515245 + h * 12345 + 12345) % 2147483648).double() / 2147483648.0   # [0,1)
            u = ((u * 7.0 + (idx % 13).double() / 13.0) % 1.0)
            if name.endswith('running_var'):
                v = 0.5 + u
            elif name.endswith('running_mean'):
                v = 0.2 * (u - 0.5)
            elif t.dim() <= 1 and name.endswith('weight'):
                v = 0.5 + u                                      # BN gamma
            elif t.dim() <= 1:
                v = 0.2 * (u - 0.5)                              # biases, scalar loss weights
            else:
                fan_in = t[0].numel()
                v = (u - 0.5) * 2.0 * math.sqrt(3.0 / fan_in)    # variance 1/fan_in
            t.copy_(v.view(t.shape).to(t.dtype))
    return module


IOU_HEADS = {'segmentation': 'segmentation_head', 'pedestrian': 'pedestrian_head'}


def prepare_heads(decoder, shifts, swap=None):
    """The IoU fixture's treatment of a deterministically filled decoder (oracle/make_golden_iou.py, applied identically
    to the reference's decoder there and to the product's in tests/test_iou_gpu.py): the class-1 row of the last 1x1
    convolution of the segmentation / pedestrian head is re-filled from an independent sequence (the plain fill makes the
    two rows nearly parallel: no contrast between the classes), the two rows (weights and biases) are exchanged where
    ``swap[head]`` says so (the logit difference changes sign: the sparse tail of its distribution becomes the positive
    class) and the class-1 bias is shifted by ``shifts[head]``."""
    import math
    with torch.no_grad():
        for key, attr in IOU_HEADS.items():
            last = getattr(decoder, attr)[-1]
            w = last.weight                                              # (2, C, 1, 1)
            n = w[1].numel()
            idx = torch.arange(n, dtype=torch.int64)
            u = ((idx * 22695477 + 977 * (len(key) + 1) + 1) % 2147483648).double() / 2147483648.0
            u = (u * 11.0 + (idx % 7).double() / 7.0) % 1.0
            w[1].copy_(((u - 0.5) * 2.0 * math.sqrt(3.0 / n)).view(w[1].shape).to(w.dtype))
            if swap and swap.get(key):
                w.copy_(w.flip(0))
                last.bias.copy_(last.bias.flip(0))
            if key in shifts:
                last.bias[1] += float(shifts[key])
    return decoder


def det_tensor(shape, seed, scale=1.0):
    """Deterministic pseudo-random tensor in [-scale, scale) (exact integer arithmetic)."""
    n = 1
    for s in shape:
        n *= s
    idx = torch.arange(n, dtype=torch.int64)
    u = ((idx * 69069 + seed * 1234567 + 1) % 2147483648).double() / 2147483648.0
    u = (u * 5.0 + (idx % 17).double() / 17.0) % 1.0
    return ((u - 0.5) * 2.0 * scale).float().view(shape)


def sample(t, n=4096):
    """Strided sample of a tensor (what the module fixtures store)."""
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // n)
    return flat[::step].contiguous()


def check_plan_structure(plan, vox):
    """The pooling plan against the voxel ids it was built from (include/stp3_hip.h, plan sections).
    ``vox``: (BT, N, D, fH, fW) ids in the reference's order.  A RUN is a maximal stretch of consecutive rows of an
    image column and depth bin with one id >= 0; its SLOT is its position in the enumeration frame, column, LAST row,
    depth bin.  Checks the row masks, the column scan, the per-voxel scan and that every voxel lists exactly the slots
    of its runs in ascending order.  Returns the total number of runs."""
    d = plan.dims
    vox = np.asarray(vox).reshape(d.BT, d.N, d.D, d.fH, d.fW)
    nxt = np.concatenate([vox[:, :, :, 1:], np.full_like(vox[:, :, :, :1], -1)], axis=3)
    valid = vox >= 0
    ends = valid & (nxt != vox)
    # (BT, N, D, fH, fW) -> (BT, col = n*fW + w, fH, D)
    order = lambda a: a.transpose(0, 1, 4, 3, 2).reshape(d.BT, d.N * d.fW, d.fH, d.D)
    ends_c, valid_c, vox_c = order(ends), order(valid), order(vox)
    bits = (np.uint64(1) << np.arange(d.D, dtype=np.uint64))
    want_masks = np.stack([(ends_c * bits).sum(-1, dtype=np.uint64), (valid_c * bits).sum(-1, dtype=np.uint64)], -1)
    got_masks = plan.masks().cpu().numpy().view(np.uint64)
    assert np.array_equal(got_masks, want_masks), 'row masks'
    per_col = ends_c.reshape(d.BT * d.N * d.fW, -1).sum(1)
    col_off = plan.column_offsets().cpu().numpy()
    assert np.array_equal(col_off, np.concatenate([[0], np.cumsum(per_col)])), 'column scan'
    slot = np.cumsum(ends_c.reshape(-1)) - 1                       # C order of (bt, col, h, d) IS the enumeration
    # run descriptors in slot order: bin | first row << 8 | last row << 16
    prv = np.concatenate([np.full_like(vox[:, :, :, :1], -1), vox[:, :, :, :-1]], axis=3)
    starts_c = order(valid & (prv != vox))
    bt_i, col_i, h_i, d_i = np.nonzero(ends_c)                     # C order = slot order
    first = np.empty(len(h_i), dtype=np.int64)
    for k, (b_, c_, h_, dd) in enumerate(zip(bt_i, col_i, h_i, d_i)):
        hh = h_
        while not starts_c[b_, c_, hh, dd]:
            hh -= 1
        first[k] = hh
    want_desc = d_i | (first << 8) | (h_i << 16)
    assert np.array_equal(plan.run_descriptors().cpu().numpy()[:len(want_desc)], want_desc), 'run descriptors'
    assert np.array_equal(plan.run_voxels().cpu().numpy()[:len(want_desc)], vox_c[bt_i, col_i, h_i, d_i]), 'run voxels'
    off = plan.offsets().cpu().numpy()
    lists = plan.run_lists().cpu().numpy()
    for bt in range(d.BT):
        e = ends_c[bt].reshape(-1)
        ids = vox_c[bt].reshape(-1)[e]
        slots = slot.reshape(d.BT, -1)[bt][e]
        assert np.array_equal(np.diff(off[bt]), np.bincount(ids, minlength=d.V)), 'voxel scan'
        frame0 = col_off[bt * d.N * d.fW]
        assert off[bt][-1] == col_off[(bt + 1) * d.N * d.fW] - frame0
        by_voxel = np.lexsort((slots, ids))                        # voxel ascending, then slot ascending
        assert np.array_equal(lists[frame0:frame0 + len(slots)], slots[by_voxel]), f'run lists of frame {bt}'
    assert int(plan.counts.abs().max()) == 0, 'count scratch not left clean'
    return int(col_off[-1])


# ----------------------------------------------------------------------------------------------
# per-block taps of a whole training step (oracle/make_golden_step.py on the reference, tests/test_step_parity_gpu.py
# on the product): input / output / output-gradient fingerprints of every block, same code on both sides
# ----------------------------------------------------------------------------------------------
BLOCK_CLASS_NAMES = ('MBConvBlock', 'BasicBlock', 'UpsamplingAdd', 'UpsamplingConcat', 'TemporalBlock', 'DeepLabHead')
DECODER_HEADS = {'segmentation': 'segmentation_head', 'pedestrian': 'pedestrian_head', 'hdmap': 'hdmap_head',
                 'instance_center': 'instance_center_head', 'instance_offset': 'instance_offset_head',
                 'instance_flow': 'instance_future_head'}


def fingerprint(t, n=512):
    """(strided sample in logical element order, [norm]) of a tensor, float64 norm."""
    t = t.detach()
    return sample(t.float(), n).cpu().numpy(), np.array([t.double().norm().item()])


class BlockTaps:
    """Forward hooks on every block (by class NAME, so that the reference's classes and the product's match alike):
    keeps the block's first tensor argument's fingerprint, the output tensor (gradient retained) and, after
    ``collect()``, the fingerprints of output and output-gradient."""

    def __init__(self, model, extra=(), forward_only=False):
        self.hooks, self.outs, self.fp = [], {}, {}
        self.forward_only = forward_only                 # taps of a no-grad run (outputs only)
        self.names = [n for n, m in model.named_modules() if type(m).__name__ in BLOCK_CLASS_NAMES]
        self.names += list(extra)
        mods = dict(model.named_modules())
        for n in self.names:
            self.hooks.append(mods[n].register_forward_hook(self._hook(n)))

    def _hook(self, name):
        def hook(mod, args, out):
            if name in self.outs or not torch.is_tensor(out) or not (out.requires_grad or self.forward_only):
                return
            if self.forward_only:
                self.fp[f'{name}/out'], self.fp[f'{name}/out_norm'] = fingerprint(out)
                self.outs[name] = None
                return
            out.retain_grad()
            self.outs[name] = out
            x = next((a for a in args if torch.is_tensor(a) and a.is_floating_point()), None)
            if x is not None:
                self.fp[f'{name}/in'], self.fp[f'{name}/in_norm'] = fingerprint(x)
        return hook

    def collect(self):
        for h in self.hooks:
            h.remove()
        for n, out in self.outs.items():
            if out is None:
                continue
            self.fp[f'{n}/out'], self.fp[f'{n}/out_norm'] = fingerprint(out)
            if out.grad is not None:
                self.fp[f'{n}/gout'], self.fp[f'{n}/gout_norm'] = fingerprint(out.grad)
        self.outs = {}
        return self.fp


def planning_inputs(cfg, batch=2):
    """Deterministic inputs of the planner fixtures (oracle/make_golden_planning.py and tests/test_planning_*.py build
    the SAME tensors): sampled trajectories that stay on the grid and a few that leave it, a cost volume that exercises
    both clamps, occupancy with an obstacle ahead of the ego vehicle, hd-map labels and logits with two lane dividers
    and a drivable corridor."""
    B, N, T = batch, cfg.PLANNING.SAMPLE_NUM, cfg.N_FUTURE_FRAMES
    step_y = (det_tensor((B, N, T), 301).abs() * 6.0)                     # 0..6 m forward per 0.5 s
    step_x = det_tensor((B, N, T), 302, 1.5)
    scale = torch.where(torch.arange(N) % 11 == 10, 4.0, 1.0).view(1, N, 1)   # every 11th leaves the 100 m grid
    y = torch.cumsum(step_y, dim=2) * scale
    x = torch.cumsum(step_x, dim=2) * scale
    trajs = torch.stack([x, y, det_tensor((B, N, T), 303)], dim=-1)
    gy = torch.cumsum(det_tensor((B, T), 304).abs() * 5.0, dim=1)
    gx = torch.cumsum(det_tensor((B, T), 305, 0.8), dim=1)
    gt = torch.stack([gx, gy, det_tensor((B, T), 306)], dim=-1)
    occ = det_tensor((B, T, 200, 200), 307) > 0.99
    occ[:, :, 112:118, 96:104] = True                                     # an obstacle 6-9 m ahead
    labels = torch.zeros(B, 2, 200, 200, dtype=torch.long)
    labels[:, 0, :, 92] = 1
    labels[:, 0, :, 108] = 1
    labels[:, 0, 140, 92:109] = 1
    labels[:, 1, :, 85:116] = 1
    labels[1, 1, 120:, :] = 0                                             # batch element 1: the road ends 10 m ahead
    logits = det_tensor((B, 4, 200, 200), 308, 1.0)
    logits[:, 1, :, 92] += 3.0
    logits[:, 1, :, 108] += 3.0
    logits[:, 3, :, 85:116] += 3.0
    logits[:, 2, :, :85] += 3.0
    logits[:, 2, :, 116:] += 3.0
    return {'trajs': trajs, 'sample_trajs': trajs, 'gt_trajs': gt, 'occupancy': occ, 'hdmap_labels': labels,
            'hdmap_logits': logits, 'cost_volume': det_tensor((B, T, 200, 200), 309, 2.0),
            'target': torch.tensor([[-3.0, 20.0], [0.0, 0.0]])[:B].contiguous(), 'commands': ['LEFT', 'FORWARD'][:B],
            'cam_front': det_tensor((B, 64, 28, 60), 310), 'w_fo': det_tensor((B, N, T), 311)}


def planning_metric_trajs(cfg, batch=2):
    """(plan, expert) (B, T, 3) for the collision counts of PlanningMetric against ``planning_inputs``' occupancy:
    sample 0 drives straight through the obstacle 6-9 m ahead while its expert passes 6 m to the side; sample 1's
    expert itself touches the obstacle at the second step (that step must not be counted)."""
    T = cfg.N_FUTURE_FRAMES
    y = torch.arange(1, T + 1, dtype=torch.float32) * 3.6
    plan = torch.zeros(batch, T, 3)
    plan[:, :, 1] = y
    plan[1, :, 0] = 0.4
    expert = plan.clone()
    expert[0, :, 0] = 6.0
    return plan, expert


def image_bytes(shape, seed):
    """Deterministic pseudo-random uint8 image batch with some low-frequency structure (exact integer arithmetic)."""
    n = 1
    for s in shape:
        n *= s
    idx = torch.arange(n, dtype=torch.int64)
    noise = ((idx * 1103515245 + seed * 12345 + 7) % 2147483648) >> 16
    *_, h, w, c = shape
    y = (idx // (w * c)) % h
    x = (idx // c) % w
    v = (noise % 97 + (x * 3 + y * 5 + (idx % c) * 40) % 160) % 256
    return v.to(torch.uint8).view(shape).numpy()
