"""CPU: the prediction stage against fixtures from the reference's own classes
(oracle/make_golden_prediction.py -> tests/golden/prediction.npz): Bottleneck, ConvNeXt Block, Bottleblock, SpatialGRU,
Dual_GRU, DistributionModule, FuturePrediction -- forward outputs, input gradients and every parameter gradient in
float32 train mode; and the state-dict keys / shapes of the reference's TrainingModule for nuscenes/Prediction.yml.
The same cases run on the MI355X in tests/test_prediction_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

PREDICTION = {'N_FUTURE_FRAMES': 4, 'PROBABILISTIC.ENABLED': True, 'PROBABILISTIC.METHOD': 'GAUSSIAN',
              'SEMANTIC_SEG.PEDESTRIAN.ENABLED': False, 'SEMANTIC_SEG.HDMAP.ENABLED': False, 'INSTANCE_FLOW.ENABLED': True,
              'INSTANCE_SEG.ENABLED': True, 'FUTURE_DISCOUNT': 0.95}


def cases():
    from stp3_amd.layers.convolutions import Block, Bottleblock, Bottleneck
    from stp3_amd.layers.temporal import Dual_GRU, SpatialGRU
    from stp3_amd.models.distributions import DistributionModule
    from stp3_amd.models.future_prediction import FuturePrediction
    return {
        'bottleneck_ds': (lambda: Bottleneck(64, 32, downsample=True), [((2, 64, 25, 31), 41)], 42),
        'block': (lambda: Block(64), [((2, 64, 20, 24), 43)], 44),
        'bottleblock': (lambda: Bottleblock(64, 32), [((2, 64, 20, 24), 45)], 46),
        'spatial_gru': (lambda: SpatialGRU(64, 64), [((1, 4, 64, 40, 40), 47), ((1, 64, 40, 40), 48)], 49),
        'dual_gru': (lambda: Dual_GRU(32, 64, n_future=3), [((1, 1, 32, 40, 40), 50), ((1, 3, 64, 40, 40), 51)], 52),
        'distribution': (lambda: DistributionModule(64, 32), [((2, 1, 64, 200, 200), 53)], 54),
        'future_prediction': (lambda: FuturePrediction(64, 32, n_future=4),
                              [((1, 1, 32, 200, 200), 55), ((1, 3, 64, 200, 200), 56)], 57),
    }


def rel(a, ref):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    r = torch.as_tensor(np.asarray(ref)).double().flatten()
    return ((a - r).norm() / r.norm().clamp_min(1e-30)).item()


def run_case(name, device='cpu', autocast=False):
    from tests.test_train_parity_gpu import make_deterministic_train
    make, inputs, seed = cases()[name]
    g = H.load('prediction.npz')
    m = make_deterministic_train(H.fill_deterministic(make())).to(device)
    ins = [H.det_tensor(shape, s).to(device).requires_grad_(True) for shape, s in inputs]
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        y = m(*ins)
    (y.float() * H.det_tensor(tuple(y.shape), seed).to(device)).sum().backward()
    errs = {'out': rel(H.sample(y.float()).cpu(), g[f'{name}/out'])}
    for i, t in enumerate(ins):
        errs[f'dx{i}'] = rel(H.sample(t.grad).cpu(), g[f'{name}/dx{i}'])
    num, den = 0.0, 0.0
    for pname, p in m.named_parameters():
        key = f'{name}/grad/{pname}'
        assert key in g.files and p.grad is not None, pname
        a, r = H.sample(p.grad, 256).double().cpu(), torch.from_numpy(g[key]).double()
        num += float((a - r).square().sum())
        den += float(r.square().sum())
    errs['dparam'] = (num / max(den, 1e-60)) ** 0.5
    return errs


@pytest.mark.parametrize('name', ['bottleneck_ds', 'block', 'bottleblock', 'spatial_gru', 'dual_gru', 'distribution'])
def test_prediction_layers_match_the_reference(name):
    errs = run_case(name)
    assert max(errs.values()) <= 2e-4, errs


def test_future_prediction_matches_the_reference():
    errs = run_case('future_prediction')
    assert errs['out'] <= 2e-4 and max(errs.values()) <= 2e-3, errs


def test_prediction_config_state_dict_matches_reference():
    from stp3_amd.config import perception_cfg
    from stp3_amd.trainer import TrainingModule
    want = json.load(open(os.path.join(H.GOLDEN, 'state_dict_keys.json')))['TrainingModule_prediction']
    got = {k: list(v.shape) for k, v in TrainingModule(perception_cfg(**PREDICTION).convert_to_dict()).state_dict().items()}
    missing, extra = sorted(set(want) - set(got)), sorted(set(got) - set(want))
    assert not missing and not extra, (missing[:5], extra[:5])
    assert not [k for k in want if want[k] != got[k]]
    assert len(want) > 1000


def test_frame_major_sequences_are_the_reference_sequences():
    """layers/temporal.py keeps (B,S,C,H,W) sequences as [S][B][H][W][C] memory between the recurrent and the per-frame stages
    of the prediction stage.  Values, indexing and gradients must be those of the reference's ``x[:, t]`` / ``torch.stack(...,
    dim=1)`` / ``x.view(b * s, c, h, w)`` (stp3/layers/temporal.py:34-40, stp3/models/future_prediction.py:33-45); only the
    strides differ: every frame is a dense channels-last tensor and the whole sequence a channels-last batch (a view)."""
    from stp3_amd.layers.temporal import batch_major, frames_as_batch, stack_frames, unbind_frames
    g = torch.Generator().manual_seed(5)
    b, s, c, h, w = 2, 3, 8, 4, 5
    frames = [torch.randn(b, c, h, w, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(s)]
    ref_frames = [f.detach().clone().requires_grad_(True) for f in frames]
    x, xr = stack_frames(frames), torch.stack(ref_frames, dim=1)
    assert x.shape == xr.shape and torch.equal(x, xr)
    assert all(f.is_contiguous(memory_format=torch.channels_last) and torch.equal(f, r)
               for f, r in zip(unbind_frames(x), xr.unbind(1)))
    x4, restore = frames_as_batch(x)
    assert x4.is_contiguous(memory_format=torch.channels_last) and x4.data_ptr() == x.data_ptr()      # a view, no copy
    # a per-frame function of the batch (frame order inside the batch is free), then back to (B,S,...)
    wgt = torch.randn(6, c, 1, 1, generator=g)
    y = restore(torch.nn.functional.conv2d(x4, wgt))
    yr = torch.nn.functional.conv2d(xr.reshape(b * s, c, h, w), wgt).reshape(b, s, 6, h, w)
    assert torch.allclose(y, yr, atol=1e-6)
    z = batch_major(y)
    assert torch.equal(z, y) and z.reshape(b * s, 6, h, w).is_contiguous(memory_format=torch.channels_last)
    # gradients: through the frame-major functions and through the reference's indexing
    probe = torch.randn(y.shape, generator=g)
    first = unbind_frames(x)[0]
    ((z * probe).sum() + (first * 0.5).sum()).backward()
    ((yr * probe).sum() + (xr[:, 0] * 0.5).sum()).backward()
    for f, r in zip(frames, ref_frames):
        assert torch.allclose(f.grad, r.grad, atol=1e-6)
