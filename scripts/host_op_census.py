"""Which torch operators the host code issues in one training step, with shapes and the repository line that issued
them -- taken on the CPU dry run of the GPU code path (tests/model_trace.py: recording stand-in library, no GPU), so it
runs in this container.  Forward operators carry their Python call site; backward ones (autograd thread) only shapes.
    python scripts/host_op_census.py [op-substring ...]        e.g.  _to_copy cat copy_ zeros"""
import collections, os, sys, tempfile, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import host_trace, model_trace

tmp = tempfile.mkdtemp()
os.environ['STP3_TRACE_LOG'] = os.path.join(tmp, 'trace.log')
os.environ.setdefault('STP3_REAL_LIB', os.path.join(ROOT, 'st-p3_amd', 'stp3_amd', 'libstp3hip.so'))
recorder = host_trace.build_recorder(os.path.join(tmp, 'rec.so'))
module, batch, cfg = model_trace.dry_setup(recorder, deterministic_fill=False)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from stp3_amd.parallel import FlatAdam, GradientBuckets

want = sys.argv[1:] or ['_to_copy', 'cat', 'copy_', 'zeros', 'fill_', 'clone', 'contiguous', 'pad']
counts = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__
        if any(w in name for w in want):
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            dt = [str(a.dtype).replace('torch.', '') for a in args if isinstance(a, torch.Tensor)][:1]
            site = ''
            for fr in reversed(traceback.extract_stack(limit=40)):
                if 'stp3_amd' in fr.filename and 'host_op_census' not in fr.filename:
                    site = f'{os.path.basename(fr.filename)}:{fr.lineno}'
                    break
            counts[(name, str(shapes), str(dt), site or '(autograd)')] += 1
        return func(*args, **(kwargs or {}))


model = module.model
buckets = GradientBuckets(model)
opt = FlatAdam(buckets, lr=1e-3, weight_decay=1e-7)
model.prepare_plan(batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'], torch.device('cpu'))


def step():
    buckets.zero_grad()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        loss = module.training_step(batch)
    loss.backward()
    opt.clip_and_step(5.0)


step()
with Census():
    step()
if os.environ.get('TOTALS') == '1':
    tot = collections.Counter()
    for (name, shapes, dt, site), n in counts.items():
        tot[name] += n
    for name, n in tot.most_common(60):
        print(f'{n:6d}  {name}')
    sys.exit(0)
for (name, shapes, dt, site), n in sorted(counts.items(), key=lambda kv: -kv[1])[:int(os.environ.get('TOP', '70'))]:
    print(f'{n:5d}  {name:28s} {dt:12s} {shapes:46s} {site}')
