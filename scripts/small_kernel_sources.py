"""Where the SMALL torch kernels of a training step come from: one step of bench.py's workload under torch.profiler with
Python stacks, every aten operator that launched a device kernel grouped by (operator, innermost frame inside this
repository), with launches per step and device time.

    python scripts/small_kernel_sources.py [--batch 4] [--top 60]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--top', type=int, default=70)
    ap.add_argument('--by-time', action='store_true', help='sort by device time instead of launches')
    ap.add_argument('--workload', default='c3', choices=sorted(bench.WORKLOADS))
    ap.add_argument('--aten-only', action='store_true', help='torch operators only (not the custom autograd functions)')
    a = ap.parse_args()
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    device = torch.device('cuda', 0)
    module, cfg = bench.build_module(device, sync_bn=True, workload=a.workload)
    buckets = GradientBuckets(module.model, gather=True)
    opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
    batch = bench.make_device_batch(a.batch, device, seed=100, workload=a.workload)

    def step():
        buckets.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = module.training_step(batch)
        loss.backward()
        buckets.finish()
        opt.clip_and_step(cfg.GRAD_NORM_CLIP)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    groups = collections.defaultdict(lambda: [0, 0.0, set()])
    for ev in prof.events():
        if not ev.kernels or ev.cpu_parent is not None and ev.cpu_parent.kernels:
            continue
        if a.aten_only and not ev.name.startswith('aten::'):
            continue                                     # leaf-most operator that owns the kernels only
        where = '?'
        for fr in ev.stack or []:
            if 'st-p3_amd' in fr or '/bench.py' in fr:
                where = fr.replace(ROOT + '/', '')
                break
        if not ev.stack:                                  # autograd thread: no Python stack; name of the backward node instead
            p = ev.cpu_parent
            while p is not None and p.cpu_parent is not None:
                p = p.cpu_parent
            where = 'backward of ' + (p.name if p is not None else '?')
        g = groups[(ev.name, where)]
        g[0] += len(ev.kernels)
        g[1] += sum(k.duration for k in ev.kernels)
        g[2].add(str(ev.input_shapes)[:80])
    rows = sorted(groups.items(), key=lambda kv: -(kv[1][1] if a.by_time else kv[1][0]))
    total = sum(v[0] for v in groups.values())
    print(f'{total} kernel launches from aten operators in one step')
    for (name, where), (n, us, shapes) in rows[:a.top]:
        print(f'{n:5d} launches {us:9.1f} us  {name:32s} {where[:110]}  {sorted(shapes)[:2]}')


if __name__ == '__main__':
    main()
