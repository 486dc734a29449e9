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
