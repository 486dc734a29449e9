"""Planner cost evaluation on the MI355X: stp3_traj_cost_fwd / _bwd (csrc/stp3_plan.hip) at the sizes of
nuscenes/Planning.yml (1 800 sampled trajectories x 6 steps) against the module's torch statements -- the reference's
formulation, ~150 tensor operators and (B, N, T, K) gathers -- on the same device.

    python scripts/time_plan.py [batch]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
from stp3_amd import synthetic  # noqa: E402
from stp3_amd.config import perception_cfg  # noqa: E402
from stp3_amd.cost import Cost_Function  # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    T, N = 6, 1800
    cfg = perception_cfg(**{'N_FUTURE_FRAMES': T, 'PLANNING.ENABLED': True, 'PLANNING.SAMPLE_NUM': N})
    p = synthetic.make_planning_inputs(B, T, N, seed=5)
    trajs = p['sample_trajectory'][:, :, 1:, :2].cuda().contiguous()
    g = torch.Generator().manual_seed(11)
    cv = (torch.randn(B, T, 200, 200, generator=g) * 0.5).cuda().requires_grad_(True)
    occ = (torch.rand(B, T, 200, 200, generator=g) < 0.02).cuda()
    hd = torch.randn(B, 4, 200, 200, generator=g).cuda()
    lane, drv = hd[:, 0:2], hd[:, 2:4]
    tgt = p['target_point'].cuda()
    w = torch.randn(B, N, T, generator=g).cuda()
    cf = Cost_Function(cfg).cuda()

    def kernel_fwd():
        with torch.no_grad():
            return cf(cv, trajs, occ, lane, drv, tgt)

    def kernel_fwd_bwd():
        cv.grad = None
        fc, fo = cf(cv, trajs, occ, lane, drv, tgt)
        (fo * w).sum().backward()

    cv64 = cv.detach().double().requires_grad_(True)
    args64 = (trajs.double(), occ, lane.double(), drv.double(), tgt.double())

    def statements_fwd():
        with torch.no_grad():
            return cf(cv64, *args64)

    def statements_fwd_bwd():
        cv64.grad = None
        fc, fo = cf(cv64, *args64)
        (fo * w.double()).sum().backward()

    k0, s0 = cf.safetycost.footprint(0).shape[0], cf.safetycost.footprint(2).shape[0]
    gathers = B * N * T * (3 * k0 + s0 + 25 + 1)
    maps_mb = B * (2 * T + 2) * 200 * 200 * 4 / 1e6
    t_k, t_kb = timed(kernel_fwd), timed(kernel_fwd_bwd)
    t_s, t_sb = timed(statements_fwd, n=5, warm=1), timed(statements_fwd_bwd, n=5, warm=1)
    print(f'B={B} N={N} T={T}: footprints {k0} / {s0} cells, {gathers / 1e6:.1f} M cell reads per call, maps {maps_mb:.1f} MB')
    print(f'  kernel    (preprocessing + stp3_traj_cost_fwd): {t_k:8.1f} us = {gathers / t_k / 1e3:.1f} G cell reads/s;'
          f'  with loss-side backward: {t_kb:8.1f} us')
    print(f'  statements (float64 tensors take the torch route, same device): fwd {t_s:8.1f} us, fwd+bwd {t_sb:8.1f} us'
          f'  -> {t_s / t_k:.1f}x / {t_sb / t_kb:.1f}x')


if __name__ == '__main__':
    main()
