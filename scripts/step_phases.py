"""Host enqueue time and GPU time of the phases of bench.py's training step, un-profiled.

    python scripts/step_phases.py [--batch 4] [--reps 8]

For each phase (label preparation + forward + losses | backward | gradient buckets + clip + Adam): `host` = wall time of
the Python call with the GPU drained before it (the call returns when its last kernel is queued), `gpu` = wall time until
the GPU has finished the phase's work when the whole phase was QUEUED beforehand is not observable from the host; what is
measured instead is `host + drain` (time from the call to the GPU being idle again).  A phase whose host time is close to
its host + drain time is host-bound: the GPU keeps up with the launches and waits for the next one.
"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--reps', type=int, default=8)
    a = ap.parse_args()
    from stp3_amd.parallel import FlatAdam, GradientBuckets
    device = torch.device('cuda', 0)
    module, cfg = bench.build_module(device, sync_bn=True)
    buckets = GradientBuckets(module.model, gather=True)
    opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
    batch = bench.make_device_batch(a.batch, device, seed=100)
    sync = torch.cuda.synchronize
    acc = {}

    def phase(name, fn):
        sync()
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        e = acc.setdefault(name, [[], []])
        e[0].append((t1 - t0) * 1e3)
        e[1].append((t2 - t0) * 1e3)
        return out

    def fwd():
        buckets.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            return module.training_step(batch)

    for i in range(a.reps + 3):
        if i == 3:
            acc.clear()
        loss = phase('forward (labels, model, losses)', fwd)
        phase('backward', loss.backward)
        phase('buckets + clip + Adam', lambda: (buckets.finish(), opt.clip_and_step(cfg.GRAD_NORM_CLIP)))
    med = lambda v: sorted(v)[len(v) // 2]
    out = {k: {'host_ms': round(med(h), 2), 'host_plus_drain_ms': round(med(t), 2)} for k, (h, t) in acc.items()}
    out['sum_host_ms'] = round(sum(v['host_ms'] for v in out.values()), 2)
    out['sum_host_plus_drain_ms'] = round(sum(v['host_plus_drain_ms'] for v in list(out.values())[:3]), 2)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
