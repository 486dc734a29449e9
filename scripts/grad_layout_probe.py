"""Which module outputs of the bench step receive a gradient that is NOT in channels-last memory?  (each such gradient
makes the additions / kernels that consume it take slow strided paths)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import torch
import bench
dev = torch.device('cuda', 0)
module, cfg = bench.build_module(dev, sync_bn=False, workload='c3')
batch = bench.make_device_batch(4, dev, seed=100, workload='c3')
seen = []


def hook_out(name):
    def fwd(mod, args, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        for k, o in enumerate(outs):
            if torch.is_tensor(o) and o.requires_grad and o.dim() == 4 and o.numel() >= 1 << 20:
                cl_out = o.is_contiguous(memory_format=torch.channels_last)
                o.register_hook(lambda g, name=name, k=k, cl_out=cl_out, shape=tuple(o.shape):
                                seen.append((name, k, shape, str(g.dtype).replace('torch.', ''), cl_out,
                                             g.is_contiguous(memory_format=torch.channels_last))))
    return fwd


for n, m in module.model.named_modules():
    m.register_forward_hook(hook_out(n))
with torch.autocast('cuda', dtype=torch.bfloat16):
    loss = module.training_step(batch)
loss.backward()
bad = [s for s in seen if not s[5]]
print(f'{len(seen)} gradients of module outputs checked, {len(bad)} not channels-last:')
for s in bad:
    print('  ', s)

# ---- second pass: every intermediate (12, 64, 200, 200) tensor of the forward, with the line that made it
import traceback
from torch.overrides import TorchFunctionMode
seen2 = []


class Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if torch.is_tensor(out) and out.requires_grad and tuple(out.shape) == (12, 64, 200, 200):
            fr = [f for f in traceback.extract_stack() if 'stp3_amd' in f.filename]
            site = f"{fr[-1].filename.split('stp3_amd/')[-1]}:{fr[-1].lineno} {getattr(func, '__name__', func)}" if fr else str(func)
            try:
                out.register_hook(lambda g, site=site: seen2.append((site, str(g.dtype).replace('torch.', ''),
                                                                   g.is_contiguous(memory_format=torch.channels_last))))
            except RuntimeError:
                pass
        return out


module.model.zero_grad(set_to_none=True)
with Mode():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = module.training_step(batch)
loss.backward()
print('intermediate (12, 64, 200, 200) tensors whose gradient is not channels-last:')
for s in seen2:
    if not s[2]:
        print('  ', s)
