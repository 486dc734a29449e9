"""One-off numerical cross-check: gradient error of the temporal model in bf16 -- the product path (hand-written
convolution / BatchNorm kernels) and plain torch bf16 autocast (vendor convolutions, torch BatchNorm arithmetic) --
both against the float32 run of the same module on the same input."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))
import copy
import torch
import torch.nn.functional as F
from stp3_amd.layers import fused, temporal, convolutions
from stp3_amd.models.temporal_model import TemporalModel
from stp3_amd.utils import to_channels_last
from tests import helpers as H


def run(m, x, gy, mode):
    m = copy.deepcopy(m)
    if mode != 'fp32':
        to_channels_last(m)
    x = x.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=mode != 'fp32'):
        y = m(x)
    (y.float() * gy).sum().backward()
    return y.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in m.named_parameters()}


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


m = H.fill_deterministic(TemporalModel(70, 3, input_shape=(200, 200), start_out_channels=64)).cuda().train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
x = H.det_tensor((1, 3, 70, 200, 200), 21).cuda()
gy = H.det_tensor((1, 3, 64, 200, 200), 22).cuda()
ref = run(m, x, gy, 'fp32')
ours = run(m, x, gy, 'bf16')
# plain torch: route every fused call to the torch statements
saved = (fused._use_mfma, temporal.bn_act, convolutions.bn_act, fused.bn_act)
fused._use_mfma = lambda *a, **k: False
for modl in (temporal, convolutions, fused):
    modl.bn_act = fused.bn_act_reference
torch_bf16 = run(m, x, gy, 'bf16')
fused._use_mfma, temporal.bn_act, convolutions.bn_act, fused.bn_act = saved
for tag, r in (('product bf16', ours), ('torch bf16  ', torch_bf16)):
    groups = {}
    for n in ref[2]:
        g = n.split('.')[0] + '.' + n.split('.')[1]
        a, b = groups.setdefault(g, ([], []))
        a.append(r[2][n].flatten()); b.append(ref[2][n].flatten())
    print(tag, 'out %.3e' % rel(r[0], ref[0]), 'dx %.3e' % rel(r[1], ref[1]),
          ' '.join(f'{g}={rel(torch.cat(a), torch.cat(b)):.3e}' for g, (a, b) in groups.items()))
