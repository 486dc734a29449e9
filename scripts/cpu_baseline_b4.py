"""The CPU baseline of bench.py (oracle/cpu_model.py: the step on plain torch CPU operators, lift by the reference's
algorithm) at B = 1 -- the bounded sample bench.py times on every run -- and ONCE at B = 4, the batch the GPU line is quoted
on: how the per-sample rate moves with the batch.  Test infrastructure (imports oracle/), never part of the product.

    python scripts/cpu_baseline_b4.py [threads]      -> one JSON line
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from oracle.cpu_model import CpuPortSTP3
    from stp3_amd import synthetic
    from stp3_amd.config import perception_cfg
    from stp3_amd.trainer import TrainingModule
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    cfg = perception_cfg(**bench.WORKLOAD_CFG['c3'])
    module = TrainingModule(cfg.convert_to_dict())
    port = CpuPortSTP3(cfg)
    port.load_state_dict(module.model.state_dict(), strict=False)
    for name in ('segmentation_weight', 'pedestrian_weight', 'hdmap_weight', 'depths_weight', 'centerness_weight',
                 'offset_weight', 'flow_weight'):
        if hasattr(module.model, name):
            setattr(port, name, getattr(module.model, name))
    module.model = port
    module.train()
    opt = module.configure_optimizers()
    out = {'threads': threads, 'cpu_model': next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?')}
    for b in (1, 4):
        batch = synthetic.make_batch(batch=b, seq=3, seed=0, gt_depth=True, instance=True)

        def step():
            t0 = time.time()
            opt.zero_grad()
            loss = module.training_step(batch)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(module.model.parameters(), cfg.GRAD_NORM_CLIP)
            opt.step()
            return time.time() - t0
        step()                                            # warm-up
        t = step()
        out[f'B{b}_s_per_step'] = round(t, 2)
        out[f'B{b}_samples_per_s'] = round(b / t, 4)
    out['B4_over_B1_rate'] = round(out['B4_samples_per_s'] / out['B1_samples_per_s'], 3)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
