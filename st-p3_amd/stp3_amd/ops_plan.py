"""Trajectory-cost evaluation of the planner on the HIP library (csrc/stp3_plan.hip): the autograd face of
``stp3_traj_cost_fwd`` / ``_bwd`` (include/stp3_hip.h), i.e. of the reference's ``Cost_Function.forward``
(stp3/cost.py:26-47).  Only the cost volume is differentiable -- trajectories, occupancy, hd map and target are data
(stp3/trainer.py:175-189 passes labels and a detached camera feature).
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check as _check


def supported(cost_volume, trajs):
    """GPU tensors, float32 or bf16 cost volume (float64 tensors take the torch statements of ``cost.py``)."""
    return cost_volume.is_cuda and trajs.is_cuda and cost_volume.dtype in (torch.float32, torch.bfloat16, torch.float16)


def _dims(params, B, N, T, H, W, K0, KL):
    d = _lib.PlanDims()
    d.B, d.N, d.T, d.H, d.W, d.K0, d.KL = B, N, T, H, W, K0, KL
    for k, v in params.items():
        setattr(d, k, float(v))
    return d


class _TrajCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost_volume, trajs, occupancy, drivable, lane, target, target_sum, fp0, fpl, params):
        B, N, T, _ = trajs.shape
        H, W = cost_volume.shape[-2:]
        d = _dims(params, B, N, T, H, W, fp0.shape[0], fpl.shape[0])
        dev = cost_volume.device
        fc = torch.empty(B, N, device=dev, dtype=torch.float32)
        fo = torch.empty(B, N, T, device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        cell = torch.empty(B, N, T, device=dev, dtype=torch.int32) if need else None
        scale = torch.empty(B, N, T, device=dev, dtype=torch.float32) if need else None
        _check(_lib.lib().stp3_traj_cost_fwd(ctypes.byref(d), ops._ptr(trajs), ops._ptr(cost_volume), ops._ptr(occupancy), ops._ptr(drivable),
                                            ops._ptr(lane), ops._ptr(target), ops._ptr(target_sum), ops._ptr(fp0), ops._ptr(fpl), ops._ptr(fc),
                                            ops._ptr(fo), ops._ptr(cell) if need else None, ops._ptr(scale) if need else None,
                                            ops._stream()), 'stp3_traj_cost_fwd')
        ctx.dims = d
        ctx.shape = cost_volume.shape
        ctx.mark_non_differentiable(fc)                 # comfort + progress depend on the trajectories only
        if need:
            ctx.save_for_backward(cell, scale)
        return fc, fo

    @staticmethod
    def backward(ctx, _g_fc, g_fo):
        cell, scale = ctx.saved_tensors
        g = torch.empty(ctx.shape, device=g_fo.device, dtype=torch.float32)
        _check(_lib.lib().stp3_traj_cost_bwd(ctypes.byref(ctx.dims), ops._ptr(g_fo.contiguous().float()), ops._ptr(cell), ops._ptr(scale),
                                            ops._ptr(g), ops._stream()), 'stp3_traj_cost_bwd')
        return (g,) + (None,) * 9


def traj_cost(cost_volume, trajs, occupancy, drivable, lane, target, fp0, fpl, params):
    """(cost_fc (B, N), cost_fo (B, N, T)) float32.  ``cost_volume`` (B, T, H, W); ``trajs`` (B, N, T, >= 2) unflipped;
    ``occupancy`` (B, T, H, W) any dtype (0 / 1); ``drivable`` / ``lane`` (B, H, W) preprocessed masks; ``target``
    (B, 2); ``fp0`` / ``fpl`` (K, 2) int32 footprint tables on the device; ``params``: the float fields of
    ``stp3_plan_dims``."""
    cv = cost_volume.float().contiguous()
    tr = trajs[..., :2].float().contiguous()
    tgt = target.float().contiguous()
    fc, fo = _TrajCost.apply(cv, tr, occupancy.float().contiguous(), drivable.float().contiguous(),
                             lane.float().contiguous(), tgt, tgt.sum().reshape(1), fp0, fpl, params)
    # always float32, whatever the cost volume's dtype: the reference indexes a half-precision cost volume and ADDS the
    # float32 terms (stp3/cost.py:36-47: the sum promotes to float32); rounding the costs to bf16 (spacing 0.25 .. 1 at
    # their magnitude of 32 .. 200) would re-rank near-tied trajectories and quantise the max-margin hinge
    return fc, fo
