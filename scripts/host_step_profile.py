"""Where does the HOST time of one training step go?  (no GPU needed)

The step is host-bound on MI355X.  This script runs bench.py's eager step -- forward, backward,
gradient buckets, clipping, FlatAdam -- as a dry run (tests/model_trace.py: do-nothing stand-in for libstp3hip.so,
tensors claim to be on the GPU) on a tiny configuration, so that what is measured is almost purely per-operator
host work: Python glue, autograd, torch dispatch.  The operator COUNT of the step does not depend on the image
size, so the breakdown carries over to the full-size step.

    python scripts/host_step_profile.py [--cpp] [--top 35] [ENV switches as for bench.py]
"""
import argparse
import cProfile
import os
import pstats
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--top', type=int, default=35)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--worker', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker is None:
        from tests import host_trace
        with tempfile.TemporaryDirectory() as tmp:
            rec = host_trace.build_recorder(os.path.join(tmp, 'librec.so'))
            env = dict(os.environ, STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=os.devnull,
                       STP3_REAL_LIB=os.path.join(ROOT, 'st-p3_amd', 'stp3_amd', 'libstp3hip.so'))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', rec, '--top', str(args.top),
                                   '--steps', str(args.steps)], env=env)
        return

    import torch
    from tests import model_trace
    module, batch, cfg = model_trace.dry_setup(args.worker, final_dim=(32, 48), batch_size=1, bev_cells=32,
                                                deterministic_fill=False)
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    buckets = GradientBuckets(module.model)
    opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
    model = module.model
    model.prepare_plan(batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'], torch.device('cpu'))

    def step():
        buckets.zero_grad()
        with torch.autocast('cpu', dtype=torch.bfloat16):
            loss = module.training_step(batch)
        loss.backward()
        buckets.finish()
        opt.clip_and_step(5.0)

    for _ in range(2):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    wall = (time.perf_counter() - t0) / args.steps
    print(f'host time per dry-run step: {wall * 1e3:.1f} ms  (tiny tensors; includes the CPU kernels of the torch ops)')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(args.top)


if __name__ == '__main__':
    main()
