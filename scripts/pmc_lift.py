"""Workload for the rocprofv3 --pmc passes: a calibration stream (known bytes) + the lift forward / backward at
the bench shape (B=4, T=3).  See scripts/gpu_pmc.sh."""
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
for _ in range(3):
    torch.add(src, 1.0, out=dst)
    bev = ops.lift_splat(f, l, plan, 0.5)
    bev.backward(torch.ones_like(bev))
torch.cuda.synchronize()
print('done')
