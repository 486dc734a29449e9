"""Which torch operators (not our kernels) cost GPU time in one bench step, with input shapes: torch.profiler over two
steady-state steps of bench.py's workload.   python scripts/torch_op_profile.py [top_n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from stp3_amd.parallel import FlatAdam, GradientBuckets

dev = torch.device('cuda', 0)
module, cfg = bench.build_module(dev, sync_bn=False, workload='c3')
buckets = GradientBuckets(module.model)
opt = FlatAdam(buckets, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY)
batch = bench.make_device_batch(4, dev, seed=100, workload='c3')


def step():
    buckets.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = module.training_step(batch)
    loss.backward()
    opt.clip_and_step(cfg.GRAD_NORM_CLIP)


for _ in range(3):
    step()
torch.cuda.synchronize()
STACKS = len(sys.argv) > 2 and sys.argv[2] == 'stacks'
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=STACKS) as prof:
    step()
    step()
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda e: -e.self_device_time_total)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if STACKS:                    # torch's own operators with the repo frames that called them (forward only: the autograd
    rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=30)   # thread has no Python stack)
            if e.key.startswith('aten::') and e.device_time_total / 2e3 > 0.03]
    rows = sorted(rows, key=lambda e: -e.device_time_total)
    for e in rows[:top]:
        frames = [f for f in e.stack if 'stp3_amd' in f or 'bench.py' in f][:4]
        print(f'{e.device_time_total / 2e3:8.3f} {e.count / 2:6.1f}  {e.key[:26]:26s} {str(e.input_shapes)[:90]}')
        for f in frames:
            print('            ', f[-110:])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == 'aten':          # only torch's own operators, by GPU time incl. children
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::')]
    rows = sorted(rows, key=lambda e: -e.device_time_total)
    print(f'{"GPU ms/step":>12s} {"calls/step":>10s}  op  shapes')
    for e in rows[:top]:
        print(f'{e.device_time_total / 2e3:12.3f} {e.count / 2:10.1f}  {e.key[:28]:28s} {str(e.input_shapes)[:170]}')
    sys.exit(0)
print(f'{"self GPU ms/step":>16s} {"calls/step":>10s}  op  shapes')
for e in rows[:top]:
    if e.self_device_time_total <= 0:
        continue
    print(f'{e.self_device_time_total / 2e3:16.3f} {e.count / 2:10.1f}  {e.key[:40]:40s} {str(e.input_shapes)[:150]}')
