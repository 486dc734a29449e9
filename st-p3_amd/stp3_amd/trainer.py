"""``TrainingModule`` -- the reference's PyTorch-Lightning module surface without the Lightning
dependency (``stp3/trainer.py:14-462``): ``TrainingModule(hparams_dict)``, ``.model``, ``.cfg``,
``shared_step(batch, is_train) -> (output, labels, loss_dict)``, ``training_step``,
``validation_step``, ``configure_optimizers``, ``prepare_future_labels``.  Parameter names match
the reference's checkpoints (``model.*`` incl. the learned uncertainty scalars
``model.<task>_weight``, trainer.py:42-97), so ``load_state_dict(ckpt['state_dict'])`` works.

Instance / flow / planning heads follow the config gates exactly as the reference does; the prediction stage
(N_FUTURE_FRAMES > 0) and the planner (PLANNING.ENABLED) are built by ``STP3`` when the configuration asks for them.
"""
import os

import torch
import torch.nn as nn

from .config import get_cfg
from .geometry import (cumulative_warp_features, cumulative_warp_features_reverse, label_warp_thetas,
                       warp_with_theta)
from .losses import DepthLoss, HDmapLoss, SegmentationLoss, SpatialRegressionLoss
from .metrics import IntersectionOverUnion, PlanningMetric
from .models.stp3 import STP3


# see TrainingModule._prepare_future_labels_batched (bit-identical to the per-label path, tests/test_host_cpu.py)
_BATCHED_LABEL_WARP = True


def _scalar():
    return nn.Parameter(torch.tensor(0.0), requires_grad=True)


def _load_checkpoint_file(path, map_location, trust_pickle):
    """torch.load of a checkpoint dictionary; safe (``weights_only=True``) unless the caller vouches for the file."""
    if trust_pickle:
        return torch.load(path, map_location=map_location, weights_only=False)
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as exc:                      # pickle.UnpicklingError of a disallowed global, mostly
        raise RuntimeError(f'{path}: not loadable with weights_only=True ({type(exc).__name__}: {exc}); if the file is a '
                           f'legacy Lightning checkpoint from a source you trust, pass trust_pickle=True') from exc


class TrainingModule(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        cfg = get_cfg(cfg_dict=hparams)
        self.cfg = cfg
        self.n_classes = len(cfg.SEMANTIC_SEG.VEHICLE.WEIGHTS)
        self.hdmap_class = cfg.SEMANTIC_SEG.HDMAP.ELEMENTS
        assert cfg.LIFT.X_BOUND[1] > 0 and cfg.LIFT.Y_BOUND[1] > 0
        self.spatial_extent = (cfg.LIFT.X_BOUND[1], cfg.LIFT.Y_BOUND[1])

        self.model = STP3(cfg)
        self.losses_fn = nn.ModuleDict()

        seg = cfg.SEMANTIC_SEG
        self.losses_fn['segmentation'] = SegmentationLoss(
            class_weights=torch.Tensor(seg.VEHICLE.WEIGHTS), use_top_k=seg.VEHICLE.USE_TOP_K,
            top_k_ratio=seg.VEHICLE.TOP_K_RATIO, future_discount=cfg.FUTURE_DISCOUNT)
        self.model.segmentation_weight = _scalar()
        self.metric_vehicle_val = IntersectionOverUnion(self.n_classes)
        if seg.PEDESTRIAN.ENABLED:
            self.losses_fn['pedestrian'] = SegmentationLoss(
                class_weights=torch.Tensor(seg.PEDESTRIAN.WEIGHTS), use_top_k=seg.PEDESTRIAN.USE_TOP_K,
                top_k_ratio=seg.PEDESTRIAN.TOP_K_RATIO, future_discount=cfg.FUTURE_DISCOUNT)
            self.model.pedestrian_weight = _scalar()
            self.metric_pedestrian_val = IntersectionOverUnion(self.n_classes)
        if seg.HDMAP.ENABLED:
            self.losses_fn['hdmap'] = HDmapLoss(
                class_weights=torch.Tensor(seg.HDMAP.WEIGHTS), training_weights=seg.HDMAP.TRAIN_WEIGHT,
                use_top_k=seg.HDMAP.USE_TOP_K, top_k_ratio=seg.HDMAP.TOP_K_RATIO)
            self.metric_hdmap_val = nn.ModuleList([IntersectionOverUnion(2, absent_score=1)
                                                   for _ in self.hdmap_class])
            self.model.hdmap_weight = _scalar()
        if cfg.LIFT.GT_DEPTH:
            self.losses_fn['depths'] = DepthLoss()
            self.model.depths_weight = _scalar()
        if cfg.INSTANCE_SEG.ENABLED:
            self.losses_fn['instance_center'] = SpatialRegressionLoss(norm=2, future_discount=cfg.FUTURE_DISCOUNT)
            self.losses_fn['instance_offset'] = SpatialRegressionLoss(
                norm=1, future_discount=cfg.FUTURE_DISCOUNT, ignore_index=cfg.DATASET.IGNORE_INDEX)
            self.model.centerness_weight = _scalar()
            self.model.offset_weight = _scalar()
        if cfg.INSTANCE_FLOW.ENABLED:
            self.losses_fn['instance_flow'] = SpatialRegressionLoss(
                norm=1, future_discount=cfg.FUTURE_DISCOUNT, ignore_index=cfg.DATASET.IGNORE_INDEX)
            self.model.flow_weight = _scalar()
        if cfg.PLANNING.ENABLED:
            self.metric_planning_val = PlanningMetric(cfg, cfg.N_FUTURE_FRAMES)
            self.model.planning_weight = _scalar()
        self.training_step_count = 0
        self._fused_terms = None

    # ------------------------------------------------------------------------------------------
    def _weighted(self, loss, name, key, value):
        """Learned-uncertainty weighting 1/(2 exp(w)) * L + w/2 (trainer.py:125-172).  With ``self._fused_terms`` set
        (``training_step``) the term is only recorded: ``_fused_total`` then weighs and adds all of them with a
        handful of operators instead of ~6 tiny ones per term in each direction."""
        w = getattr(self.model, f'{name}_weight')
        if self._fused_terms is not None:
            self._fused_terms.append((value, w))
            return
        loss[key] = value / (2 * torch.exp(w))
        loss[f'{name}_uncertainty'] = 0.5 * w

    @staticmethod
    def _fused_total(terms):
        """sum_k L_k / (2 exp(w_k)) + w_k / 2 over the recorded (L_k, w_k)."""
        values = torch.stack([v.reshape(()) for v, _ in terms])
        w = torch.stack([p.reshape(()) for _, p in terms]).to(values.dtype)
        return (values * (0.5 * torch.exp(-w)) + 0.5 * w).sum()

    def shared_step(self, batch, is_train, fused_total=False):
        """``fused_total``: return ``{'total': sum of all weighted terms}`` instead of the per-term dictionary (what
        ``training_step`` needs; the terms themselves are for logging)."""
        self._fused_terms = [] if (fused_total and is_train) else None
        labels = self.prepare_future_labels(batch)
        output = self.model(batch['image'], batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'])
        rf = self.model.receptive_field
        cfg = self.cfg
        loss = {}
        if is_train:
            self._weighted(loss, 'segmentation', 'segmentation',
                           self.losses_fn['segmentation'](output['segmentation'], labels['segmentation'], rf))
            if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
                self._weighted(loss, 'pedestrian', 'pedestrian',
                               self.losses_fn['pedestrian'](output['pedestrian'], labels['pedestrian'], rf))
            if cfg.SEMANTIC_SEG.HDMAP.ENABLED:
                self._weighted(loss, 'hdmap', 'hdmap', self.losses_fn['hdmap'](output['hdmap'], labels['hdmap']))
            if cfg.INSTANCE_SEG.ENABLED:
                self._weighted(loss, 'centerness', 'instance_center', self.losses_fn['instance_center'](
                    output['instance_center'], labels['centerness'], rf))
                self._weighted(loss, 'offset', 'instance_offset', self.losses_fn['instance_offset'](
                    output['instance_offset'], labels['offset'], rf))
            if cfg.LIFT.GT_DEPTH:
                self._weighted(loss, 'depths', 'depths',
                               self.losses_fn['depths'](output['depth_prediction'], labels['depths']))
            if cfg.INSTANCE_FLOW.ENABLED:
                self._weighted(loss, 'flow', 'instance_flow', self.losses_fn['instance_flow'](
                    output['instance_flow'], labels['flow'], rf))
            final_traj = None
            if cfg.PLANNING.ENABLED:
                # trainer.py:175-193: occupancy and hd map are LABELS in training, the camera feature is detached
                occupancy = labels['segmentation'][:, rf:].squeeze(2).bool()
                if 'pedestrian' in labels:
                    occupancy = occupancy | labels['pedestrian'][:, rf:].squeeze(2).bool()
                pl_loss, final_traj = self.model.planning(
                    cam_front=output['cam_front'].detach(), trajs=batch['sample_trajectory'][:, :, 1:],
                    gt_trajs=labels['gt_trajectory'][:, 1:], cost_volume=output['costvolume'][:, rf:],
                    semantic_pred=occupancy, hd_map=labels['hdmap'], commands=batch['command'],
                    target_points=batch['target_point'])
                self._weighted(loss, 'planning', 'planning', pl_loss)
            if self._fused_terms is not None:
                loss['total'] = self._fused_total(self._fused_terms)
                self._fused_terms = None
            output = {**output, 'selected_traj': self._with_origin(final_traj, labels['gt_trajectory'])}
        else:
            # evaluate.py:95-98 / trainer.py:216-236: argmax over classes, present frame onwards
            seg_pred = torch.argmax(output['segmentation'].detach(), dim=2, keepdim=True)
            self.metric_vehicle_val(seg_pred[:, rf - 1:], labels['segmentation'][:, rf - 1:])
            if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
                ped_pred = torch.argmax(output['pedestrian'].detach(), dim=2, keepdim=True)
                self.metric_pedestrian_val(ped_pred[:, rf - 1:], labels['pedestrian'][:, rf - 1:])
            if cfg.SEMANTIC_SEG.HDMAP.ENABLED:
                for i in range(len(self.hdmap_class)):
                    hd_pred = torch.argmax(output['hdmap'][:, 2 * i:2 * (i + 1)].detach(), dim=1, keepdim=True)
                    self.metric_hdmap_val[i](hd_pred, labels['hdmap'][:, i:i + 1])
            final_traj = None
            if cfg.PLANNING.ENABLED:
                # trainer.py:230-246: predicted occupancy and hd map steer the planner, the labels judge the result
                occupancy = seg_pred.bool()
                if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
                    occupancy = occupancy | ped_pred.bool()
                _, final_traj = self.model.planning(
                    cam_front=output['cam_front'].detach(), trajs=batch['sample_trajectory'][:, :, 1:],
                    gt_trajs=labels['gt_trajectory'][:, 1:], cost_volume=output['costvolume'][:, rf:].detach(),
                    semantic_pred=occupancy[:, rf:].squeeze(2), hd_map=output['hdmap'].detach(),
                    commands=batch['command'], target_points=batch['target_point'])
                truth = labels['segmentation'][:, rf:].squeeze(2).bool()
                if 'pedestrian' in labels:
                    truth = truth | labels['pedestrian'][:, rf:].squeeze(2).bool()
                self.metric_planning_val(final_traj, labels['gt_trajectory'][:, 1:], truth)
            output = {**output, 'selected_traj': self._with_origin(final_traj, labels['gt_trajectory'])}
        return output, labels, loss

    @staticmethod
    def _with_origin(final_traj, gt_trajectory):
        """The planned trajectory with the ego origin in front (trainer.py:192-193), or the expert's without a planner."""
        if final_traj is None:
            return gt_trajectory
        return torch.cat([torch.zeros_like(final_traj[:, :1]), final_traj], dim=1)

    def _warp_pair(self, x, ego, rf, to_long):
        """Past frames warped into the present frame, future frames warped back (trainer.py:279-290)."""
        past = cumulative_warp_features(x[:, :rf].float(), ego[:, :rf], mode='nearest',
                                        spatial_extent=self.spatial_extent)
        fut = cumulative_warp_features_reverse(x[:, rf - 1:].float(), ego[:, rf - 1:], mode='nearest',
                                               spatial_extent=self.spatial_extent)
        if to_long:
            past, fut = past.long(), fut.long()
        return torch.cat([past.contiguous()[:, :-1], fut.contiguous()], dim=1)

    def _prepare_future_labels_batched(self, batch):
        """``prepare_future_labels`` with the pose chains evaluated once per batch (on the host, where the pose
        tensors live) and ALL label maps warped together: one ``grid_sample`` per past / future frame over the
        channel concatenation instead of one per label type and frame, each preceded by its own ~40 tiny pose
        operators.  Nearest sampling acts per channel, so the result equals the per-label warps bit for bit."""
        cfg, rf = self.cfg, self.model.receptive_field
        dev = batch['segmentation'].device
        labels = {'hdmap': batch['hdmap'][:, rf - 1].long().contiguous(), 'gt_trajectory': batch['gt_trajectory']}
        if cfg.LIFT.GT_DEPTH:
            ds = self.model.encoder_downsample
            depth = batch['depths'][:, :rf, :, ::ds, ::ds]
            depth = torch.clamp(depth, cfg.LIFT.D_BOUND[0], cfg.LIFT.D_BOUND[1] - 1) - cfg.LIFT.D_BOUND[0]
            labels['depths'] = depth.long().contiguous()
        items = [('segmentation', batch['segmentation'], True)]
        if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
            items.append(('pedestrian', batch['pedestrian'], True))
        if cfg.INSTANCE_SEG.ENABLED:
            items += [('instance', batch['instance'].unsqueeze(2), True), ('centerness', batch['centerness'], False),
                      ('offset', batch['offset'], False)]
        if cfg.INSTANCE_FLOW.ENABLED:
            items.append(('flow', batch['flow'], False))
        stacked = torch.cat([x.float() for _, x, _ in items], dim=2)                  # (B,S,sum C,H,W)
        prepared = batch.get('_prepared')
        if prepared is not None:
            # the warp matrices of this batch already sit in device memory (``prepare_batch``): no host work, no upload --
            # what a captured step needs (stp3_amd/graph.py), and the same launch as below
            from . import ops_loss
            b_, s_ = stacked.shape[:2]
            warped = ops_loss.warp_nearest(stacked.reshape(b_ * s_, *stacked.shape[2:]), prepared['warp_theta'],
                                           prepared['warp_ident']).view(stacked.shape)
        else:
            warped = self._warp_labels(stacked, batch['future_egomotion'], rf, dev)
        c0 = 0
        for name, x, to_long in items:
            part = warped[:, :, c0:c0 + x.shape[2]]
            c0 += x.shape[2]
            labels[name] = (part.long() if to_long else part).contiguous()
        if 'instance' in labels:
            labels['instance'] = labels['instance'][:, :, 0]
        return labels

    def _label_warp_matrices(self, future_egomotion, b_, s_, rf):
        """(theta (B*S, 2, 3) float32, identity flags (B*S) int32), on the host: the matrices of ``label_warp_thetas`` frame
        by frame, identity (flag 1: the frame is copied) where a frame is not warped."""
        thetas = label_warp_thetas(future_egomotion.detach().float().cpu(), rf, self.spatial_extent)
        eye = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
        th = torch.stack([thetas[t].float().cpu() if t in thetas else eye.expand(b_, 2, 3) for t in range(s_)], dim=1)
        ident = torch.tensor([0 if t in thetas else 1 for _ in range(b_) for t in range(s_)], dtype=torch.int32)
        return th.reshape(b_ * s_, 2, 3).contiguous(), ident

    def _warp_labels(self, stacked, future_egomotion, rf, dev):
        b_, s_ = stacked.shape[:2]
        if stacked.is_cuda:
            # all frames and all label channels in ONE launch (stp3_warp_nearest): frames without a theta are copied
            from . import ops_loss
            th, ident = self._label_warp_matrices(future_egomotion, b_, s_, rf)
            return ops_loss.warp_nearest(stacked.reshape(b_ * s_, *stacked.shape[2:]), th, ident).view(stacked.shape)
        thetas = label_warp_thetas(future_egomotion.detach().float().cpu(), rf, self.spatial_extent)
        frames = []
        for t in range(stacked.shape[1]):
            frame = stacked[:, t]
            if t in thetas:
                frame = warp_with_theta(frame, thetas[t].to(dev), mode='nearest')
            frames.append(frame)
        return torch.stack(frames, dim=1)

    def prepare_batch(self, batch, device, out=None):
        """Everything of a training step that depends on the batch's POSES and is computed on the host, done ahead of the
        step and left in device memory: the label-warp matrices (``_label_warp_matrices``), the ego-motion vectors the
        temporal model reads, and the voxel-pool plan (``STP3.prepare_plan``).  With ``batch['_prepared']`` set to the
        result, ``training_step`` touches no host data and uploads nothing -- the step can be captured into a hipGraph and
        replayed (stp3_amd/graph.py).  ``out``: a previous result whose device buffers are overwritten in place."""
        rf = self.model.receptive_field
        b_, s_ = batch['segmentation'].shape[:2]
        th, ident = self._label_warp_matrices(batch['future_egomotion'], b_, s_, rf)
        ego = batch['future_egomotion'].detach().float().cpu()
        if out is None:
            out = {'warp_theta': th.to(device), 'warp_ident': ident.to(device), 'ego': ego.to(device), 'plan': None}
        else:
            if '_upload' not in out:
                from . import ops
                out['_upload'] = {k: ops.PinnedUpload(out[k]) for k in ('warp_theta', 'warp_ident', 'ego')}
            for k, v in (('warp_theta', th), ('warp_ident', ident), ('ego', ego)):
                out['_upload'][k](v)                     # (asynchronous: pinned double buffers)
        out['plan'] = self.model.prepare_plan(batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'], device,
                                              out=out['plan'])
        self.model.prebuilt_plan = None          # (the caller decides which forward uses it: ``STP3.prebuilt_plan``)
        return out

    def prepare_future_labels(self, batch):
        if _BATCHED_LABEL_WARP:
            return self._prepare_future_labels_batched(batch)
        cfg, rf = self.cfg, self.model.receptive_field
        dev = batch['segmentation'].device
        ego = batch['future_egomotion'].to(dev)
        labels = {'hdmap': batch['hdmap'][:, rf - 1].long().contiguous(), 'gt_trajectory': batch['gt_trajectory']}
        if cfg.LIFT.GT_DEPTH:
            ds = self.model.encoder_downsample
            depth = batch['depths'][:, :rf, :, ::ds, ::ds]
            depth = torch.clamp(depth, cfg.LIFT.D_BOUND[0], cfg.LIFT.D_BOUND[1] - 1) - cfg.LIFT.D_BOUND[0]
            labels['depths'] = depth.long().contiguous()
        labels['segmentation'] = self._warp_pair(batch['segmentation'], ego, rf, True)
        if cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED:
            labels['pedestrian'] = self._warp_pair(batch['pedestrian'], ego, rf, True)
        if cfg.INSTANCE_SEG.ENABLED:
            labels['instance'] = self._warp_pair(batch['instance'].unsqueeze(2), ego, rf, True)[:, :, 0]
            labels['centerness'] = self._warp_pair(batch['centerness'], ego, rf, False)
            labels['offset'] = self._warp_pair(batch['offset'], ego, rf, False)
        if cfg.INSTANCE_FLOW.ENABLED:
            labels['flow'] = self._warp_pair(batch['flow'], ego, rf, False)
        return labels

    def training_step(self, batch, batch_idx=0):
        _, _, loss = self.shared_step(batch, True, fused_total=True)
        self.training_step_count += 1
        return loss['total']

    def validation_step(self, batch, batch_idx=0):
        output, labels, _ = self.shared_step(batch, False)
        return {'step_val_seg_iou_dynamic': self.metric_vehicle_val.compute()[1]}

    # ------------------------------------------------------------------------------------------
    # checkpoint surface: what train.py:21-29 and evaluate.py:31 call on the reference's module
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location='cpu', strict=True, trust_pickle=False, **overrides):
        """``TrainingModule.load_from_checkpoint(path, strict=True)`` of the Lightning surface (evaluate.py:31): the
        checkpoint is the dictionary Lightning writes -- ``hyper_parameters`` (the config as a plain dict,
        trainer.py:19) and ``state_dict`` (``model.*`` keys incl. the learned loss weights) -- and the module is
        rebuilt from the former before the latter is loaded.  ``overrides`` replace top-level hyper-parameters.
        The file is read with ``weights_only=True`` (tensors, plain containers and scalars: what ``checkpoint()`` writes);
        ``trust_pickle=True`` runs the full pickle machinery instead, for legacy Lightning files that carry other
        objects -- only for files you trust, unpickling executes code."""
        ckpt = _load_checkpoint_file(checkpoint_path, map_location, trust_pickle)
        hparams = ckpt.get('hyper_parameters', ckpt.get('hparams'))
        if hparams is None:
            raise KeyError(f'{checkpoint_path}: no "hyper_parameters" entry (not a Lightning checkpoint of TrainingModule)')
        if isinstance(hparams, dict) and set(hparams) == {'hparams'}:        # saved through save_hyperparameters()
            hparams = hparams['hparams']
        hparams = {**dict(hparams), **overrides}
        module = cls(hparams)
        module.load_state_dict(ckpt['state_dict'], strict=strict)
        return module

    def load_pretrained_weights(self, path, map_location='cpu', trust_pickle=False):
        """train.py:21-29 (``PRETRAINED.LOAD_WEIGHTS``): initialise from a single-image model -- every tensor of the
        checkpoint's ``state_dict`` whose key exists here and does not belong to a decoder, non-strict.  Returns the
        keys that were loaded."""
        weights = _load_checkpoint_file(path, map_location, trust_pickle)['state_dict']
        state = self.state_dict()
        weights = {k: v for k, v in weights.items() if k in state and 'decoder' not in k}
        self.load_state_dict(weights, strict=False)
        from . import ops
        if any(p.is_cuda for p in self.parameters()):
            ops.invalidate_weight_cache()                # bf16 shadows of the convolution weights follow the new values
        return sorted(weights)

    def checkpoint(self):
        """The dictionary ``load_from_checkpoint`` reads (Lightning's layout, minus optimizer / loop state)."""
        from . import ops
        ops.flush_batch_counters()
        return {'hyper_parameters': dict(self.hparams), 'state_dict': self.state_dict()}

    def configure_optimizers(self):
        return torch.optim.Adam(self.model.parameters(), lr=self.cfg.OPTIMIZER.LR,
                                weight_decay=self.cfg.OPTIMIZER.WEIGHT_DECAY)
