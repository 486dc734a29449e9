"""Synthetic nuScenes-shaped inputs for the LSS camera->BEV path.

There is no dataset in this environment, so tests, ``bench.py`` and the golden
generator all draw their inputs from here.  The batch *schema* follows what the
reference's ``TrainingModule.shared_step`` consumes (reference
``stp3/trainer.py:102-108, 257-347``; produced by
``stp3/datas/NuscenesData.py:569-646``):

    image            (B, S, N, 3, H, W) float32   ImageNet-normalised pixels
    intrinsics       (B, S, N, 3, 3)    float32   upper-triangular K
    extrinsics       (B, S, N, 4, 4)    float32   camera -> ego
    future_egomotion (B, S, 6)          float32   (tx,ty,tz,rx,ry,rz) frame t -> t+1
    segmentation / pedestrian (B, S, 1, X, Y) int64, hdmap (B, S, 2, X, Y) int64
    gt_trajectory, command, sample_trajectory, target_point (unused on this path)

The rig is a nuScenes-like 6-camera ring: yaw
{55, 0, -55, 110, 180, -110} deg, cameras 1.5 m from the origin at z = 1.5 m,
intrinsics scaled by the reference's resize 0.3 / top-crop 46
(``stp3/config.py:63-64``), with small random roll/pitch/yaw so that points do
not sit exactly on voxel borders (generic case).  ``axis_aligned=True`` removes
the jitter: the border-degenerate stress case (CARLA's rig is axis aligned,
``stp3/datas/CarlaData.py:310-313``).

Everything is generated on the CPU with an explicit ``torch.Generator`` so the
same seed gives the same bits on every box.
"""
import math

import torch

CAM_YAWS_DEG = (55.0, 0.0, -55.0, 110.0, 180.0, -110.0)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float64)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]], dtype=torch.float64)


# camera frame (z forward, x right, y down) -> ego frame (x forward, y left, z up)
_CAM_TO_EGO = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]], dtype=torch.float64)


def make_rig(batch, seq, n_cams=6, final_dim=(224, 480), seed=0, axis_aligned=False,
             resize_scale=0.3, top_crop=46.0):
    """Return (intrinsics (B,S,N,3,3), extrinsics (B,S,N,4,4), future_egomotion (B,S,6))."""
    g = torch.Generator().manual_seed(seed)
    h, w = final_dim
    # 224x480 corresponds to the reference's scale 0.3 / crop 46 of 900x1600; other
    # resolutions keep the same field of view by scaling with the width.
    scale = resize_scale * (w / 480.0)
    fx = 1266.0 * scale
    cx = 816.0 * scale
    cy = 491.0 * scale - top_crop * (h / 224.0)

    intr = torch.zeros(batch, seq, n_cams, 3, 3, dtype=torch.float64)
    extr = torch.zeros(batch, seq, n_cams, 4, 4, dtype=torch.float64)
    for b in range(batch):
        for s in range(seq):
            for n in range(n_cams):
                jit = torch.zeros(3) if axis_aligned else torch.randn(3, generator=g) * 3.0
                k = torch.tensor([[fx + jit[0].item(), 0.0, cx + jit[1].item()],
                                  [0.0, fx + jit[0].item(), cy + jit[2].item()],
                                  [0.0, 0.0, 1.0]], dtype=torch.float64)
                intr[b, s, n] = k
                yaw = math.radians(CAM_YAWS_DEG[n % len(CAM_YAWS_DEG)])
                if axis_aligned:
                    # snap the ring to the axes: 0 / +-90 / 180 deg
                    yaw = round(yaw / (math.pi / 2)) * (math.pi / 2)
                    rpy = torch.zeros(3)
                    dt = torch.zeros(3)
                else:
                    rpy = torch.randn(3, generator=g) * 0.03
                    dt = torch.randn(3, generator=g) * 0.05
                rot = _rot_z(yaw + rpy[2].item()) @ _rot_y(rpy[1].item()) @ _rot_x(rpy[0].item()) @ _CAM_TO_EGO
                pos = torch.tensor([1.5 * math.cos(yaw), 1.5 * math.sin(yaw), 1.5], dtype=torch.float64)
                extr[b, s, n, :3, :3] = rot
                extr[b, s, n, :3, 3] = pos + dt.double()
                extr[b, s, n, 3, 3] = 1.0

    ego = torch.zeros(batch, seq, 6, dtype=torch.float64)
    ego[..., 0] = 2.0      # 4 m/s * 0.5 s between keyframes
    ego[..., 5] = 0.02     # gentle left turn
    if not axis_aligned:
        ego += torch.randn(batch, seq, 6, generator=g).double() * 0.01
    else:
        ego[..., 5] = 0.0
    return intr.float(), extr.float(), ego.float()


def make_labels(batch, seq, bev=(200, 200), seed=0, n_hdmap=2):
    """Bernoulli-blob occupancy labels with the reference's dtypes/shapes."""
    g = torch.Generator().manual_seed(seed + 7919)
    x, y = bev

    def blobs(p):
        coarse = (torch.rand(batch, seq, 1, (x + 3) // 4, (y + 3) // 4, generator=g) < p).float()
        up = torch.nn.functional.interpolate(coarse.view(batch * seq, 1, *coarse.shape[-2:]), scale_factor=4,
                                             mode='nearest')[..., :x, :y]
        return up.view(batch, seq, 1, x, y).long()

    seg = blobs(0.03)
    ped = blobs(0.01)
    hd = torch.cat([blobs(0.05) for _ in range(n_hdmap)], dim=2)
    return seg, ped, hd


def make_instance_labels(segmentation, ignore_index=255, seed=0):
    """Instance-branch labels derived from the occupancy blobs, with the shapes / dtypes / ignore convention the
    reference's data loader produces (stp3/datas/NuscenesData.py:518-560, stp3/utils/instance.py:11-69):
    instance (B,T,X,Y) int64 ids (0 = background; every 4x4 blob is one instance), centerness (B,T,1,X,Y) float in
    [0,1] peaking at the instance centre, offset (B,T,2,X,Y) float = vector to the centre, ``ignore_index`` outside
    instances, flow (B,T,2,X,Y) float = a per-sample constant motion inside instances, ``ignore_index`` outside."""
    b, t, _, x, y = segmentation.shape
    occ = segmentation[:, :, 0] > 0
    ii = torch.arange(x).view(1, 1, x, 1).expand(b, t, x, y)
    jj = torch.arange(y).view(1, 1, 1, y).expand(b, t, x, y)
    cell = (ii // 4) * ((y + 3) // 4) + jj // 4 + 1
    instance = torch.where(occ, cell, torch.zeros_like(cell)).long()
    di = 1.5 - (ii % 4).float()
    dj = 1.5 - (jj % 4).float()
    centerness = torch.where(occ, torch.exp(-(di * di + dj * dj) / 8.0), torch.zeros(())).unsqueeze(2)
    ign = torch.full((), float(ignore_index))
    offset = torch.stack([torch.where(occ, di, ign), torch.where(occ, dj, ign)], dim=2)
    g = torch.Generator().manual_seed(seed + 15485863)
    motion = torch.randn(b, 1, 2, 1, 1, generator=g)
    flow = torch.where(occ.unsqueeze(2), motion.expand(b, t, 2, x, y), ign)
    return instance, centerness.float(), offset.float(), flow.float()


def make_planning_inputs(batch, n_future, sample_num, seed=0):
    """Planner inputs with the shapes of the reference's loader (stp3/datas/NuscenesData.py:589-646): the expert
    trajectory (B, n_future + 1, 3) starting at the origin, ``sample_num`` candidate trajectories (B, N, n_future + 1,
    3) -- one third per navigation command, constant-curvature arcs at 0.5 s steps -- a command and a target point per
    sample."""
    g = torch.Generator().manual_seed(seed + 32452843)
    t = torch.arange(n_future + 1, dtype=torch.float32) * 0.5
    third = sample_num // 3
    speed = 2.0 + 10.0 * torch.rand(batch, sample_num, 1, generator=g)
    turn = torch.cat([0.02 + 0.1 * torch.rand(batch, third, 1, generator=g),           # LEFT third
                      0.02 * (torch.rand(batch, third, 1, generator=g) - 0.5),          # FORWARD third
                      -0.02 - 0.1 * torch.rand(batch, sample_num - 2 * third, 1, generator=g)], dim=1)
    dist = speed * t
    heading = turn * dist
    samples = torch.stack([-dist * torch.sin(heading * 0.5), dist * torch.cos(heading * 0.5), heading], dim=-1)
    expert = samples[:, third + third // 2].clone() * (0.8 + 0.4 * torch.rand(batch, 1, 1, generator=g))
    return {'command': ['FORWARD'] * batch, 'sample_trajectory': samples.float(), 'gt_trajectory': expert.float(),
            'target_point': expert[:, -1, :2].clone().float() * 2.0}


def make_batch(batch=1, seq=3, n_cams=6, final_dim=(224, 480), bev=(200, 200), seed=0,
               axis_aligned=False, with_images=True, with_labels=True, gt_depth=False, instance=False,
               planning=None):
    """Full batch dict with the reference's keys (trainer.py:102-108).  ``planning`` = (n_future, sample_num) fills
    the planner's inputs; without it they are the placeholders of the perception path."""
    g = torch.Generator().manual_seed(seed + 104729)
    intr, extr, ego = make_rig(batch, seq, n_cams, final_dim, seed, axis_aligned)
    out = {
        'intrinsics': intr,
        'extrinsics': extr,
        'future_egomotion': ego,
        'command': ['FORWARD'] * batch,
        'sample_trajectory': torch.zeros(batch, 1, 1, 3),
        'target_point': torch.zeros(batch, 2),
        'gt_trajectory': torch.zeros(batch, 1, 3),
    }
    if planning is not None:
        out.update(make_planning_inputs(batch, planning[0], planning[1], seed))
    if with_images:
        out['image'] = torch.randn(batch, seq, n_cams, 3, *final_dim, generator=g)
    if with_labels:
        seg, ped, hd = make_labels(batch, seq, bev, seed)
        out['segmentation'] = seg
        out['pedestrian'] = ped
        out['hdmap'] = hd
        if instance:
            out['instance'], out['centerness'], out['offset'], out['flow'] = make_instance_labels(seg, seed=seed)
    if gt_depth:
        out['depths'] = torch.randint(0, 61, (batch, seq, n_cams, *final_dim), generator=g).float()
    return out


def lift_case(batch=4, seq=3, n_cams=6, final_dim=(224, 480), downsample=8, d_bound=(2.0, 50.0, 1.0),
              x_bound=(-50.0, 50.0, 0.5), y_bound=(-50.0, 50.0, 0.5), z_bound=(-10.0, 10.0, 20.0), channels=64, seed=31):
    """Inputs of the voxel pool alone, for kernel micro-benchmarks and counter runs (scripts/time_lift.py,
    scripts/pmc_lift.py): ((frustum, bev_resolution, bev_start, bev_dimension), intrinsics, extrinsics,
    future_egomotion, feat (B,S,N,C,fH,fW), depth_logits (B,S,N,D,fH,fW)).  Product-side constants only: the
    frustum of ``STP3.create_frustum`` (stp3.py:111-130) and ``models.stp3.bev_parameters``."""
    from .models.stp3 import bev_parameters
    h, w = final_dim
    fh, fw = h // downsample, w // downsample
    depth = torch.arange(*d_bound, dtype=torch.float)
    n_d = depth.shape[0]
    frustum = torch.stack((torch.linspace(0, w - 1, fw, dtype=torch.float).view(1, 1, fw).expand(n_d, fh, fw),
                           torch.linspace(0, h - 1, fh, dtype=torch.float).view(1, fh, 1).expand(n_d, fh, fw),
                           depth.view(n_d, 1, 1).expand(n_d, fh, fw)), -1)
    res, start, dim = bev_parameters(list(x_bound), list(y_bound), list(z_bound))
    intr, extr, ego = make_rig(batch, seq, n_cams, final_dim, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    feat = torch.relu(torch.randn(batch, seq, n_cams, channels, fh, fw, generator=g))
    logits = torch.randn(batch, seq, n_cams, n_d, fh, fw, generator=g) * 2.0
    return (frustum, res, start, dim), intr, extr, ego, feat, logits
