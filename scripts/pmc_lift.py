"""Workload for the rocprofv3 passes over the voxel-pool kernels: a calibration stream (known bytes) + plan build +
forward / backward at the bench shape (B=4, T=3), BEV layout and type of the model path under autocast (channels-last,
bf16 out, bf16 gradient back).  See scripts/gpu_pmc_lift.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from stp3_amd import ops
from stp3_amd import synthetic

(frustum, res, start, dim), intr, extr, ego, feat, logits = synthetic.lift_case(batch=4, seq=3, seed=31)
grid = ops.LiftGrid(frustum, res, start, dim, 'cuda')
plan = ops.LiftPlan.build(grid, intr, extr, ego, 64)
f = feat.cuda().requires_grad_(True)
l = logits.cuda().requires_grad_(True)
src = torch.randn(64 << 20, device='cuda')          # calibration: reads 256 MiB, writes 256 MiB
dst = torch.empty_like(src)
g = None
for _ in range(4):
    torch.add(src, 1.0, out=dst)
    ops.LiftPlan.build(grid, intr, extr, ego, 64, out=plan)
    bev = ops.lift_splat(f, l, plan, 0.5, True, torch.bfloat16)
    if g is None:
        g = torch.randn(bev.shape[0], bev.shape[1], bev.shape[3], bev.shape[4], bev.shape[2], device='cuda').to(torch.bfloat16).permute(0, 1, 4, 2, 3)
    bev.backward(g)
torch.cuda.synchronize()
print('done')
