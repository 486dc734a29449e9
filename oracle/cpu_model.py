"""TEST INFRASTRUCTURE -- CPU port of the whole perception step, used ONLY as bench.py's
``cpu_baseline`` (kind "port") and by tests as a checker.

The reference's own classes cannot travel to the GPU box (they live under /root/reference), so the
CPU baseline is this port: the same torch modules on the CPU, with the lift / voxel-pool done the
way the reference does it (stp3/models/stp3.py:215-301 + stp3/utils/geometry.py:299-330):
materialised depth (x) feature outer product, boolean mask, argsort, prefix-sum ``VoxelsSumming``
autograd function, per-(b,t) Python loops.  The reference-side arithmetic here is the literal
restatement validated bitwise against the reference by oracle/make_golden.py.
"""
import torch

from oracle import lift_oracle as lo
from stp3_amd.models.stp3 import STP3


class _VoxelsSumming(torch.autograd.Function):
    """geometry.py:299-330: cumsum, keep the last row of every run of equal ranks, adjacent difference."""

    @staticmethod
    def forward(ctx, x, ranks):
        x = x.cumsum(0)
        mask = torch.ones(x.shape[0], dtype=torch.bool)
        mask[:-1] = ranks[1:] != ranks[:-1]
        x = x[mask]
        x = torch.cat((x[:1], x[1:] - x[:-1]))
        ctx.save_for_backward(mask)
        return x, ranks[mask]

    @staticmethod
    def backward(ctx, grad_x, _):
        (mask,) = ctx.saved_tensors
        idx = torch.cumsum(mask, 0)
        idx[mask] -= 1
        return grad_x[idx], None


class CpuPortSTP3(STP3):
    """``STP3`` whose BEV lifting follows the reference's CPU algorithm (differentiable)."""

    def calculate_birds_eye_view_features(self, image, intrinsics, extrinsics, future_egomotion):
        b, s, n, c, h, w = image.shape
        cfg = self.cfg
        vox = lo.lift_voxel_ids(self.frustum.data, intrinsics, extrinsics, future_egomotion,
                                cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
        vox = torch.from_numpy(vox).long()
        feat, depth = self.encoder(image.reshape(b * s * n, c, h, w))
        prob = depth.softmax(dim=1)
        x = prob.unsqueeze(1) * feat.unsqueeze(2)                      # (BSN, C, D, fH, fW)  stp3.py:216
        x = x.view(b, s, n, *x.shape[1:]).permute(0, 1, 2, 4, 5, 6, 3)  # (B,S,N,D,fH,fW,C)
        xd, yd = int(self.bev_dimension[0]), int(self.bev_dimension[1])
        cc = x.shape[-1]
        frames = []
        for bi in range(b):
            bev = torch.zeros(xd * yd, cc)
            for t in range(s):
                x_b = x[bi, t].reshape(-1, cc)
                ranks = vox[bi, t].reshape(-1)
                keep = ranks >= 0
                x_b, ranks = x_b[keep], ranks[keep]
                order = ranks.argsort()
                x_b, ranks = x_b[order], ranks[order]
                tmp = torch.zeros(xd * yd, cc)
                if x_b.shape[0]:
                    sums, kept = _VoxelsSumming.apply(x_b, ranks)
                    # float32 scatter target and accumulation (stp3.py:280-296), whatever the point matrix's type: a
                    # float64 evaluation (tests/test_step_truth_cpu.py) rounds the pooled sums here exactly like the
                    # float64 reference fixture does (oracle/make_golden_step.py: to_float64)
                    tmp = tmp.index_put((kept,), sums.to(tmp.dtype))
                bev = bev * self.discount + tmp
                frames.append(bev.t().reshape(cc, xd, yd))
        out = torch.stack(frames).view(b, s, cc, xd, yd).to(feat.dtype)
        return out, depth.view(b, s, n, *depth.shape[1:]), None
