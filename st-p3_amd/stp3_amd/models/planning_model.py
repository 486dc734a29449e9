"""The planner -- drop-in for the reference's ``stp3/models/planning_model.py`` (same attribute names: ``cost_function``,
``reduce_channel``, ``GRU``, ``decoder``; same ``forward`` signature and return value).

  1. the N sampled trajectories of the navigation command are scored by ``Cost_Function`` (one HIP launch,
     csrc/stp3_plan.hip) and the cheapest is selected;
  2. training adds the max-margin loss against the expert trajectory (one more launch with N = 1);
  3. a GRU cell, started from the front camera's features (four stride-2 / channel-reducing ``Bottleneck``s: the
     library's MFMA convolutions), refines the selected trajectory point by point towards the target.
The GRU cell and the two linear layers are torch operators on (B, 256) tensors: library GEMMs, a few microseconds each.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..cost import Cost_Function
from ..layers.convolutions import Bottleneck


_AXIS_WEIGHT = {}


def _axis_weight(device, dtype):
    """The (10, 1) weights of the refinement loss (planning_model.py:146-149), uploaded once per device: a constant built
    inside the step is a host-to-device copy per step -- and not capturable into a hipGraph."""
    key = (str(device), dtype)
    t = _AXIS_WEIGHT.get(key)
    if t is None:
        t = _AXIS_WEIGHT[key] = torch.tensor([10., 1.], device=device, dtype=dtype)
    return t


class Planning(nn.Module):
    def __init__(self, cfg, feature_channel, gru_input_size=6, gru_state_size=256):
        super().__init__()
        self.cost_function = Cost_Function(cfg)
        self.sample_num = cfg.PLANNING.SAMPLE_NUM
        self.commands = cfg.PLANNING.COMMAND
        assert self.sample_num % 3 == 0
        self.num = self.sample_num // 3
        c = feature_channel
        self.reduce_channel = nn.Sequential(
            Bottleneck(c, c, downsample=True),
            Bottleneck(c, c // 2, downsample=True),
            Bottleneck(c // 2, c // 2, downsample=True),
            Bottleneck(c // 2, c // 8))
        self.GRU = nn.GRUCell(gru_input_size, gru_state_size)
        self.decoder = nn.Sequential(nn.Linear(gru_state_size, gru_state_size), nn.ReLU(inplace=True),
                                     nn.Linear(gru_state_size, 2))

    @staticmethod
    def compute_L2(trajs, gt_traj):
        """Squared planar distance, (B, N, T) or (B, T) (planning_model.py:35-46)."""
        if trajs.ndim != gt_traj.ndim or trajs.ndim not in (3, 4):
            raise ValueError('trajs ndim != gt_traj ndim')
        return ((trajs[..., :2] - gt_traj[..., :2]) ** 2).sum(dim=-1)

    def _costs(self, trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points):
        return self.cost_function(cost_volume, trajs[..., :2], semantic_pred, lane_divider, drivable_area, target_points)

    def select(self, trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points, k=1):
        """The k cheapest of the sampled trajectories (k = 1: (B, T, 3))   (planning_model.py:43-63)."""
        fc, fo = self._costs(trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points)
        best = torch.topk(fc + fo.sum(dim=-1), k, dim=-1, largest=False).indices
        return trajs[torch.arange(len(trajs), device=trajs.device)[:, None], best].squeeze(1)

    def loss(self, trajs, gt_trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points):
        """Max-margin: the expert must be cheaper than every sample by the sample's distance to it (:65-88)."""
        fc, fo = self._costs(trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points)
        if gt_trajs.ndim == 3:
            gt_trajs = gt_trajs[:, None]
        gfc, gfo = self._costs(gt_trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points)
        margin = F.relu(gfo - fo).sum(-1) + (gfc - fc) + self.compute_L2(trajs, gt_trajs).mean(dim=-1)
        return F.relu(margin).max(dim=-1).values.mean()

    def command_samples(self, trajs, commands):
        """The third of the samples that belongs to each element's command, repeated three times (:103-115)."""
        thirds = {'LEFT': 0, 'FORWARD': 1, 'RIGHT': 2}
        picked = []
        for traj, command in zip(trajs, commands):
            if command in thirds:
                k = thirds[command]
                picked.append(traj[k * self.num:(k + 1) * self.num].repeat(3, 1, 1))
            else:
                picked.append(traj)
        return torch.stack(picked)

    def forward(self, cam_front, trajs, gt_trajs, cost_volume, semantic_pred, hd_map, commands, target_points):
        """cam_front (B, C, fH, fW); trajs (B, N, T, 3); gt_trajs (B, T, 3); cost_volume / semantic_pred (B, T, H, W);
        hd_map (B, 2 | 4, H, W); commands: list of B strings; target_points (B, 2)  ->  (loss, trajectory (B, T, 3))"""
        samples = self.command_samples(trajs, commands)
        if hd_map.shape[1] == 2:
            lane_divider, drivable_area = hd_map[:, 0:1], hd_map[:, 1:2]
        elif hd_map.shape[1] == 4:
            lane_divider, drivable_area = hd_map[:, 0:2], hd_map[:, 2:4]
        else:
            raise NotImplementedError
        loss = 0
        if self.training:
            loss = self.loss(samples, gt_trajs, cost_volume, semantic_pred, lane_divider, drivable_area, target_points)
        h = self.reduce_channel(cam_front).flatten(start_dim=1)
        chosen = self.select(samples, cost_volume, semantic_pred, lane_divider, drivable_area, target_points)
        h = h.float()
        target = target_points.to(dtype=h.dtype)
        point = torch.zeros(h.shape[0], 2, device=h.device, dtype=h.dtype)
        refined = []
        for i in range(chosen.shape[1]):
            h = self.GRU(torch.cat([point, chosen[:, i, :2].to(h.dtype), target], dim=-1), h)
            point = self.decoder(h)
            refined.append(point)
        refined = torch.stack(refined, dim=1)
        refined = torch.cat([refined, torch.zeros_like(refined[..., :1])], dim=-1)
        if self.training:
            axis_weight = _axis_weight(refined.device, refined.dtype)
            loss = loss * 0.5 + (F.smooth_l1_loss(refined[..., :2], gt_trajs[..., :2].to(refined.dtype), reduction='none') *
                                 axis_weight).mean()
        return loss, refined
