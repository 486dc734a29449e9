"""Training losses of the perception path, following ``stp3/losses.py``: SpatialRegressionLoss
(:6-40), SegmentationLoss (:43-83), HDmapLoss (:85-114), DepthLoss (:116-134).

The reference ranks the per-pixel losses with a full descending sort and keeps the first k
(:76-81, :108-111); only the *mean of the k largest* is used, so ``torch.topk`` computes the same
quantity without ordering all 40 000 pixels.

GPU float32 / bf16 tensors run every loss on the kernels of csrc/stp3_loss.hip (``ops_loss``: per-pixel loss, radix
select of the k-th largest value and the top-k sum in three launches; one launch backward); CPU tensors and float64
(the parity tests' noise-free evaluation) take the torch statements below."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops_loss
from .utils import hp, staged_mean


_ROW_SCALE = {}


def _row_scale(future_discount, seq_len, n_present, batch, device):
    """(batch * seq_len,) float32 future-discount factor of every (sample, frame) row, built once per configuration."""
    key = (float(future_discount), seq_len, n_present, batch, str(device))
    t = _ROW_SCALE.get(key)
    if t is None:
        one = torch.zeros((), dtype=torch.float32)
        t = _future_discounts(future_discount, seq_len, n_present, one).repeat(batch).to(device)
        _ROW_SCALE[key] = t
    return t


def _future_discounts(future_discount, seq_len, n_present, like):
    fut = future_discount ** torch.arange(1, seq_len - n_present + 1, device=like.device, dtype=like.dtype)
    return torch.cat([torch.ones(n_present, device=like.device, dtype=like.dtype), fut], dim=0)


class SpatialRegressionLoss(nn.Module):
    def __init__(self, norm, ignore_index=255, future_discount=1.0):
        super().__init__()
        if norm not in (1, 2):
            raise ValueError(f'Expected norm 1 or 2, but got norm={norm}')
        self.norm, self.ignore_index, self.future_discount = norm, ignore_index, future_discount
        self.loss_fn = F.l1_loss if norm == 1 else F.mse_loss

    def forward(self, prediction, target, n_present=3):
        assert prediction.dim() == 5, 'Must be a 5D tensor'
        if ops_loss.supported(prediction) and target.is_cuda:
            seq_len = prediction.shape[1]
            assert seq_len >= n_present
            scale = _row_scale(self.future_discount, seq_len, n_present, prediction.shape[0], prediction.device)
            return ops_loss.regression_loss(prediction, target, scale, self.norm, float(self.ignore_index))
        mask = target[:, :, :1] != self.ignore_index
        prediction = hp(prediction)
        loss = self.loss_fn(prediction, target.to(prediction.dtype), reduction='none').sum(dim=-3, keepdim=True)
        seq_len = loss.shape[1]
        assert seq_len >= n_present
        loss = loss * _future_discounts(self.future_discount, seq_len, n_present, loss).view(1, seq_len, 1, 1, 1)
        # mean over the unmasked pixels, 0 when there are none (losses.py:31-33, :43) -- as a masked sum: boolean
        # indexing and the emptiness test would each cost a device -> host synchronisation per call
        # (torch.where, not a product: a non-finite prediction at an IGNORED pixel must not reach the sum -- inf * 0 is
        # NaN, while the reference's loss[mask] drops the pixel)
        count = mask.sum()
        return torch.where(mask, loss, loss.new_zeros(())).sum() / count.clamp_min(1).to(loss.dtype)


class SegmentationLoss(nn.Module):
    def __init__(self, class_weights, ignore_index=255, use_top_k=False, top_k_ratio=1.0, future_discount=1.0):
        super().__init__()
        # a (non-persistent) buffer: moves with the module, so the step does no host->device copy
        self.register_buffer('class_weights', torch.as_tensor(class_weights, dtype=torch.float32), persistent=False)
        self.ignore_index, self.use_top_k, self.top_k_ratio = ignore_index, use_top_k, top_k_ratio
        self.future_discount = future_discount

    def forward(self, prediction, target, n_present=3):
        if target.shape[-3] != 1:
            raise ValueError('segmentation label must be an index-label with channel dimension = 1.')
        b, s, c, h, w = prediction.shape
        assert s >= n_present
        if ops_loss.supported(prediction) and target.is_cuda:
            scale = _row_scale(self.future_discount, s, n_present, b, prediction.device)
            k = int(self.top_k_ratio * h * w) if self.use_top_k else 0
            return ops_loss.ce_topk_mean(prediction, target.reshape(b, s, h, w), self.class_weights, scale, k, self.ignore_index)
        loss = F.cross_entropy(hp(prediction.reshape(b * s, c, h, w)), target.reshape(b * s, h, w),
                               ignore_index=self.ignore_index, reduction='none',
                               weight=self.class_weights.to(device=prediction.device, dtype=hp(prediction).dtype))
        loss = loss.view(b, s, h, w)
        assert s >= n_present
        loss = loss * _future_discounts(self.future_discount, s, n_present, loss).view(1, s, 1, 1)
        loss = loss.view(b, s, -1)
        if self.use_top_k:
            k = int(self.top_k_ratio * loss.shape[2])
            loss = loss.topk(k, dim=2, sorted=False).values
        return staged_mean(loss)


class HDmapLoss(nn.Module):
    def __init__(self, class_weights, training_weights, use_top_k, top_k_ratio, ignore_index=255):
        super().__init__()
        self.register_buffer('class_weights', torch.as_tensor(class_weights, dtype=torch.float32), persistent=False)
        self.training_weights = training_weights
        self.ignore_index, self.use_top_k, self.top_k_ratio = ignore_index, use_top_k, top_k_ratio

    def forward(self, prediction, target):
        total = 0
        # (the two logits of every map element through ONE split: its backward is one concatenation, that of a slice per
        # element a zero-fill, a copy and an addition each)
        pieces = prediction.split(2, dim=1)
        for i in range(target.shape[-3]):
            cur = target[:, i]
            b = cur.shape[0]
            if ops_loss.supported(prediction) and target.is_cuda:
                k = int(self.top_k_ratio[i] * cur.shape[1] * cur.shape[2]) if self.use_top_k[i] else 0
                total = total + ops_loss.ce_topk_mean(pieces[i], cur, self.class_weights[i], None, k,
                                                      self.ignore_index) * self.training_weights[i]
                continue
            loss = F.cross_entropy(hp(pieces[i]), cur, ignore_index=self.ignore_index,
                                   reduction='none',
                                   weight=self.class_weights[i].to(device=target.device, dtype=hp(prediction).dtype)).view(b, -1)
            if self.use_top_k[i]:
                k = int(self.top_k_ratio[i] * loss.shape[1])
                loss = loss.topk(k, dim=1, sorted=False).values
            total = total + staged_mean(loss) * self.training_weights[i]
        return total


class DepthLoss(nn.Module):
    def __init__(self, class_weights=None, ignore_index=255):
        super().__init__()
        self.class_weights, self.ignore_index = class_weights, ignore_index

    def forward(self, prediction, target):
        b, s, n, d, h, w = prediction.shape
        if ops_loss.supported(prediction) and target.is_cuda:
            return ops_loss.ce_topk_mean(prediction.reshape(b * s * n, d, h, w), target.reshape(b * s * n, h, w),
                                         self.class_weights, None, 0, self.ignore_index)
        loss = F.cross_entropy(hp(prediction.reshape(b * s * n, d, h, w)), target.reshape(b * s * n, h, w),
                               ignore_index=self.ignore_index, reduction='none', weight=self.class_weights)
        return staged_mean(loss)
