"""Intersection-over-union with the reference's protocol (``stp3/metrics.py:15-71``: per-class
tp / fp / fn accumulated over batches, IoU = tp / (tp + fp + fn), ``absent_score`` when a class
has neither support nor predictions).  The reference builds on pytorch-lightning's ``Metric`` /
``stat_scores_multiple_classes`` (not installed); states are plain buffers here and ``sync()``
sums them over the process group (``dist_reduce_fx='sum'``)."""
import torch
import torch.nn as nn


class IntersectionOverUnion(nn.Module):
    def __init__(self, n_classes, ignore_index=None, absent_score=0.0, reduction='none'):
        super().__init__()
        self.n_classes, self.ignore_index, self.absent_score, self.reduction = (n_classes, ignore_index,
                                                                                 absent_score, reduction)
        for name in ('true_positive', 'false_positive', 'false_negative', 'support'):
            self.register_buffer(name, torch.zeros(n_classes), persistent=False)

    def reset(self):
        for name in ('true_positive', 'false_positive', 'false_negative', 'support'):
            getattr(self, name).zero_()

    @torch.no_grad()
    def update(self, prediction, target):
        pred, tgt = prediction.reshape(-1).long(), target.reshape(-1).long()
        n = self.n_classes
        # stat_scores_multiple_classes (metrics.py:38): per-class counts; labels outside [0, n) (an ignore value such
        # as 255) contribute to no class of their own but still make the other side's class a false positive / negative
        in_p, in_t = (pred >= 0) & (pred < n), (tgt >= 0) & (tgt < n)
        both = in_p & in_t
        conf = torch.bincount(tgt[both] * n + pred[both], minlength=n * n).view(n, n).to(self.true_positive.dtype)
        tp = conf.diag()
        stray_p = torch.bincount(pred[in_p & ~in_t], minlength=n).to(tp.dtype)
        stray_t = torch.bincount(tgt[in_t & ~in_p], minlength=n).to(tp.dtype)
        self.true_positive += tp
        self.false_positive += conf.sum(0) - tp + stray_p
        self.false_negative += conf.sum(1) - tp + stray_t
        self.support += conf.sum(1) + stray_t

    def forward(self, prediction, target):
        self.update(prediction, target)

    def sync(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            packed = torch.stack([self.true_positive, self.false_positive, self.false_negative, self.support])
            dist.all_reduce(packed, group=group)
            self.true_positive, self.false_positive, self.false_negative, self.support = packed.unbind(0)

    def compute(self):
        tp, fp, fn, sup = self.true_positive, self.false_positive, self.false_negative, self.support
        denom = tp + fp + fn
        scores = torch.where(sup + tp + fp == 0, torch.full_like(tp, self.absent_score),
                             tp / denom.clamp(min=1)).float()
        if self.ignore_index is not None and 0 <= self.ignore_index < self.n_classes:
            scores = torch.cat([scores[:self.ignore_index], scores[self.ignore_index + 1:]])
        if self.reduction == 'elementwise_mean':
            return scores.mean()
        if self.reduction == 'sum':
            return scores.sum()
        return scores


class PlanningMetric(nn.Module):
    """L2 error and collision rates of the planned trajectory per future step (``stp3/metrics.py:263-399``): ``obj_col``
    counts steps whose trajectory POINT lies in an occupied cell, ``obj_box_col`` steps whose ego BOX (the 32 cells of
    ``cost.BaseCost.footprint``) touches one; steps at which the expert's own box already collides with the labels are
    not counted.  Plain buffers + ``sync()`` instead of Lightning's metric states; evaluated for the whole batch at
    once, on the device, instead of a Python loop over samples and steps with host round trips."""

    def __init__(self, cfg, n_future=4):
        super().__init__()
        from .cost import BaseCost
        base = BaseCost(cfg)
        self.dx = nn.Parameter(base.dx.detach().clone(), requires_grad=False)
        self.bx = nn.Parameter(base.bx.detach().clone(), requires_grad=False)
        self.bev_dimension = [int(v) for v in base.bev_dimension]
        self.W, self.H = cfg.EGO.WIDTH, cfg.EGO.HEIGHT
        self.n_future = n_future
        self.register_buffer('footprint', torch.from_numpy(base.footprint(0)), persistent=False)
        for name in ('obj_col', 'obj_box_col', 'L2'):
            self.register_buffer(name, torch.zeros(n_future), persistent=False)
        self.register_buffer('total', torch.tensor(0), persistent=False)

    def reset(self):
        for name in ('obj_col', 'obj_box_col', 'L2', 'total'):
            getattr(self, name).zero_()

    def box_collisions(self, trajs, segmentation):
        """(B, T) bool: does the ego box at step t of the (flipped) trajectory (B, T, 2) touch an occupied cell."""
        B, T, _ = trajs.shape
        rc = self.footprint.to(trajs.device)
        rows = (trajs[..., 1:2] / self.dx[0] + rc[:, 0]).to(torch.int32).clamp(0, self.bev_dimension[0] - 1).long()
        cols = (trajs[..., 0:1] / self.dx[1] + rc[:, 1]).to(torch.int32).clamp(0, self.bev_dimension[1] - 1).long()
        bi = torch.arange(B, device=trajs.device).view(B, 1, 1)
        ti = torch.arange(T, device=trajs.device).view(1, T, 1)
        return segmentation[bi, ti, rows, cols].bool().any(dim=-1)

    @torch.no_grad()
    def update(self, trajs, gt_trajs, segmentation):
        """trajs, gt_trajs (B, T, 3); segmentation (B, T, H, W)."""
        assert trajs.shape == gt_trajs.shape
        self.L2 += torch.sqrt(((trajs[..., :2] - gt_trajs[..., :2]) ** 2).sum(dim=-1)).sum(dim=0)
        flip = torch.tensor([-1, 1], device=trajs.device, dtype=trajs.dtype)
        plan, expert = trajs[..., :2] * flip, gt_trajs[..., :2] * flip
        clean = ~self.box_collisions(expert, segmentation)
        yi = ((plan[..., 1] - self.bx[0]) / self.dx[0]).long()
        xi = ((plan[..., 0] - self.bx[1]) / self.dx[1]).long()
        inside = (yi >= 0) & (yi < self.bev_dimension[0]) & (xi >= 0) & (xi < self.bev_dimension[1])
        B, T = yi.shape
        bi = torch.arange(B, device=trajs.device).view(B, 1)
        ti = torch.arange(T, device=trajs.device).view(1, T)
        hit = segmentation[bi, ti, yi.clamp(0, self.bev_dimension[0] - 1), xi.clamp(0, self.bev_dimension[1] - 1)]
        self.obj_col += (hit.long() * (inside & clean)).sum(dim=0).to(self.obj_col.dtype)
        self.obj_box_col += (self.box_collisions(plan, segmentation) & clean).sum(dim=0).to(self.obj_box_col.dtype)
        self.total += B

    def forward(self, trajs, gt_trajs, segmentation):
        self.update(trajs, gt_trajs, segmentation)

    def sync(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            packed = torch.cat([self.obj_col, self.obj_box_col, self.L2, self.total.reshape(1).to(self.L2.dtype)])
            dist.all_reduce(packed, group=group)
            n = self.n_future
            self.obj_col, self.obj_box_col, self.L2 = packed[:n].clone(), packed[n:2 * n].clone(), packed[2 * n:3 * n].clone()
            self.total = packed[-1].round().to(self.total.dtype)

    def compute(self):
        return {'obj_col': self.obj_col / self.total, 'obj_box_col': self.obj_box_col / self.total, 'L2': self.L2 / self.total}
