"""CPU: activation recomputation of the EfficientNet trunk (``Encoder.recompute_blocks``: every MBConv block keeps only its
input and re-runs its forward in the backward pass -- what lets BASELINE configs[4]'s four samples per GPU of 896 x 1600
images fit one MI355X) changes NOTHING: outputs, every parameter gradient, the BatchNorm running statistics and the batch
counters are bit-equal to the plain run (the second forward runs with momentum 0 and does not count)."""
import torch
import torch.nn as nn

from tests import helpers as H


def run(recompute, device='cpu', autocast=False):
    from stp3_amd import ops
    from stp3_amd.config import perception_cfg
    from stp3_amd.models.encoder import Encoder
    m = H.fill_deterministic(Encoder(perception_cfg().MODEL.ENCODER, D=48)).train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m.backbone._global_params.drop_connect_rate = 0.0
    m.recompute_blocks = recompute
    m = m.to(device)
    x = H.det_tensor((2, 3, 64, 96), 3).to(device)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        f, d = m(x)
    (f.float().sum() + (d.float() * d.float()).sum()).backward()
    ops.flush_batch_counters()
    return (f.detach(), d.detach(), {n: p.grad.clone() for n, p in m.named_parameters()},
            {n: b.clone() for n, b in m.named_buffers()})


def check(device='cpu', autocast=False):
    a, b = run(False, device, autocast), run(True, device, autocast)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(a[2][n], b[2][n]) for n in a[2]), [n for n in a[2] if not torch.equal(a[2][n], b[2][n])][:4]
    assert all(torch.equal(a[3][n], b[3][n]) for n in a[3]), [n for n in a[3] if not torch.equal(a[3][n], b[3][n])][:4]
    assert any(n.endswith('num_batches_tracked') and int(v) == 1 for n, v in b[3].items())


def test_recomputed_trunk_equals_the_plain_one():
    check()
