"""Pose helpers and label warping used by the training step (torch ops; off the HIP hot path).

Restates the pieces of ``stp3/utils/geometry.py`` the perception path needs: ``mat2pose_vec``
(:97-121), ``euler2mat`` / ``pose_vec2mat`` (:124-172), ``invert_pose_matrix`` (:175-193),
``warp_features`` (:196-238) and the two cumulative warps (:241-296), which
``TrainingModule.prepare_future_labels`` applies to the label maps (nearest sampling).
``VoxelsSumming`` (:299-330) is re-exported from ``stp3_amd.ops`` (HIP operator, GPU only)."""
import torch
import torch.nn.functional as F

from .ops import VoxelsSumming  # noqa: F401  (same import path as the reference: stp3.utils.geometry)


def euler2mat(angle):
    """(..., 3) XYZ Euler angles -> (..., 3, 3): R = X(rx) . Y(ry) . Z(rz)."""
    shape = angle.shape
    a = angle.reshape(-1, 3)
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    zero, one = torch.zeros_like(z), torch.ones_like(z)
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    zmat = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).view(-1, 3, 3)
    ymat = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).view(-1, 3, 3)
    xmat = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).view(-1, 3, 3)
    return xmat.bmm(ymat).bmm(zmat).view(*shape[:-1], 3, 3)


def pose_vec2mat(vec):
    """(..., 6) = (tx,ty,tz,rx,ry,rz) -> (..., 4, 4)."""
    rot = euler2mat(vec[..., 3:].contiguous())
    top = torch.cat([rot, vec[..., :3].unsqueeze(-1)], dim=-1)
    bottom = torch.zeros_like(top[..., :1, :])
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], dim=-2)


def mat2pose_vec(matrix):
    rotx = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    cosy = torch.sqrt(matrix[..., 1, 2] ** 2 + matrix[..., 2, 2] ** 2)
    roty = torch.atan2(matrix[..., 0, 2], cosy)
    rotz = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.cat((matrix[..., :3, 3], torch.stack((rotx, roty, rotz), dim=-1)), dim=-1)


def invert_pose_matrix(x):
    assert x.dim() == 3 and x.shape[1:] == (4, 4)
    rt = x[:, :3, :3].transpose(1, 2)
    inv = torch.cat([rt, -torch.bmm(rt, x[:, :3, 3:])], dim=-1)
    bottom = torch.zeros_like(inv[:, :1])
    bottom[:, 0, 3] = 1.0
    return torch.cat([inv, bottom], dim=1)


def warp_theta(flow, spatial_extent):
    """(b,6) flow -> the (b,2,3) affine matrix ``warp_features`` samples with."""
    b = flow.shape[0]
    angle = flow[:, 5]
    t0 = -flow[:, 0] / spatial_extent[0]       # forward axis is inverted
    t1 = flow[:, 1] / spatial_extent[1]
    c, s = torch.cos(angle), torch.sin(angle)
    return torch.stack([c, -s, t1, s, c, t0], dim=-1).view(b, 2, 3)


def warp_with_theta(x, theta, mode='nearest'):
    grid = F.affine_grid(theta, size=x.shape, align_corners=False).to(x.dtype)
    return F.grid_sample(x, grid, mode=mode, padding_mode='zeros', align_corners=False)


def warp_features(x, flow, mode='nearest', spatial_extent=None):
    """In-plane rigid warp of a BEV map (b,c,h,w) by the xy / yaw part of a 6-DoF flow (b,6)."""
    if flow is None:
        return x
    return warp_with_theta(x, warp_theta(flow, spatial_extent), mode)


def label_warp_thetas(flow, receptive_field, spatial_extent):
    """The affine matrices ``cumulative_warp_features(x[:, :rf], flow[:, :rf])`` and
    ``cumulative_warp_features_reverse(x[:, rf-1:], flow[:, rf-1:])`` apply, computed ONCE for a batch:
    {frame index: (b,2,3) theta}; the present frame rf-1 is not warped and has no entry.  ``flow`` (b,s,6)."""
    rf, seq = receptive_field, flow.shape[1]
    thetas = {}
    mats = pose_vec2mat(flow)
    if rf > 1:
        cum = mats[:, rf - 2]
        for t in reversed(range(rf - 1)):
            thetas[t] = warp_theta(mat2pose_vec(cum), spatial_extent)
            cum = mats[:, t - 1] @ cum
    cum = None
    for i in range(1, seq - rf + 1):
        inv = invert_pose_matrix(mats[:, rf - 1 + i - 1])
        cum = inv if cum is None else cum @ inv
        thetas[rf - 1 + i] = warp_theta(mat2pose_vec(cum), spatial_extent)
    return thetas


def cumulative_warp_features(x, flow, mode='nearest', spatial_extent=None):
    """x[:, -1] unchanged; x[:, t] warped by flow[t] @ ... @ flow[-2] (past -> present)."""
    seq = x.shape[1]
    if seq == 1:
        return x
    flow = pose_vec2mat(flow)
    out = [x[:, -1]]
    cum = flow[:, -2]
    for t in reversed(range(seq - 1)):
        out.append(warp_features(x[:, t], mat2pose_vec(cum), mode=mode, spatial_extent=spatial_extent))
        cum = flow[:, t - 1] @ cum
    return torch.stack(out[::-1], 1)


def cumulative_warp_features_reverse(x, flow, mode='nearest', spatial_extent=None):
    """x[:, 0] unchanged; x[:, i] warped by flow[0]^-1 @ ... @ flow[i-1]^-1 (future -> present)."""
    flow = pose_vec2mat(flow)
    out = [x[:, 0]]
    cum = None
    for i in range(1, x.shape[1]):
        inv = invert_pose_matrix(flow[:, i - 1])
        cum = inv if cum is None else cum @ inv
        out.append(warp_features(x[:, i], mat2pose_vec(cum), mode, spatial_extent=spatial_extent))
    return torch.stack(out, 1)
