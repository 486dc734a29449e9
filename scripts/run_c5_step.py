"""BASELINE configs[4] geometry through the WHOLE model on one MI355X: 896 x 1600 images (fH x fW = 112 x 200), D = 64 depth
bins, 400 x 400 BEV grid at 0.25 m, T = 5 frames, the losses of configs[2] -- one sample per GPU (the configuration's
per-GPU batch of 4 is 720 images of 1.4 MP).  bf16 autocast, the same eager step as bench.py.  Prints ms per step and the
peak memory.

    python scripts/run_c5_step.py [batch] [steps] [recompute]

``recompute`` (third argument, 1 / 0; default: on from three samples per GPU): the MBConv blocks of the trunk keep only their
inputs and are re-run in the backward pass (Encoder.recompute_blocks) -- with it the configuration's four samples per GPU
fit the 288 GB of one MI355X.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))


def main():
    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    from stp3_amd.parallel import FlatAdam, GradientBuckets, convert_sync_batchnorm
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    recompute = bool(int(sys.argv[3])) if len(sys.argv) > 3 else B >= 3
    T = 5
    small = os.environ.get('C5_SMALL') == '1'             # T = 5 at the configs[2] image / grid size: a quick check of the path
    dim, bev, xb, db = ((224, 480), (200, 200), [-50.0, 50.0, 0.5], [2.0, 50.0, 1.0]) if small else \
        ((896, 1600), (400, 400), [-50.0, 50.0, 0.25], [2.0, 66.0, 1.0])
    torch.cuda.set_per_process_memory_fraction(0.92)          # a clean out-of-memory error, never a dead box
    cfg = perception_cfg(**{'IMAGE.FINAL_DIM': dim, 'LIFT.X_BOUND': xb, 'LIFT.Y_BOUND': xb,
                            'LIFT.D_BOUND': db, 'TIME_RECEPTIVE_FIELD': T, 'LIFT.GT_DEPTH': True,
                            'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True})
    torch.manual_seed(1234)
    module = to_channels_last(convert_sync_batchnorm(TrainingModule(cfg.convert_to_dict()), enabled=False).cuda())
    module.train()
    module.model.encoder.recompute_blocks = recompute
    batch = synthetic.make_batch(batch=B, seq=T, final_dim=dim, bev=bev, seed=0, gt_depth=True, instance=True)
    batch = {k: (v.cuda() if torch.is_tensor(v) and k not in ('intrinsics', 'extrinsics', 'future_egomotion') else v)
             for k, v in batch.items()}
    buckets = GradientBuckets(module.model, gather=True)
    opt = FlatAdam(buckets, lr=1e-3, weight_decay=1e-7)

    def step():
        buckets.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = module.training_step(batch)
        loss.backward()
        buckets.finish()
        opt.clip_and_step(5.0)
        return loss

    loss = step()
    torch.cuda.synchronize()
    print(f'first step done: loss {float(loss):.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    n_img = B * T * 6
    print(f'configs[4] geometry, B={B} per GPU, T={T}, trunk recomputation {"on" if recompute else "off"}: {ms:.1f} ms per step = {B / ms * 1e3:.2f} samples/s '
          f'({n_img} images of {dim[0]}x{dim[1]} = {n_img * dim[0] * dim[1] / (72 * 224 * 480):.1f}x the pixels of a configs[2] step); '
          f'loss {float(loss):.4f}, finite {bool(torch.isfinite(loss))}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')


if __name__ == '__main__':
    main()
