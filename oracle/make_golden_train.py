"""TEST INFRASTRUCTURE (build container only) -- TRAIN-mode forward + backward fixtures of the benchmarked path.

    python oracle/make_golden_train.py            -> tests/golden/train.npz, tests/golden/state_dict_keys.json

Runs the REFERENCE's own classes from /root/reference on the CPU in float32, in ``train()`` mode (BatchNorm on
batch statistics) with the two stochastic pieces switched off on purpose (ASPP ``Dropout`` p = 0, EfficientNet
drop-connect rate 0 -- SURVEY.md section 7, hard part 6), and records forward outputs, loss values and
gradient samples of:

  * ``TemporalModel``                      (stp3/models/temporal_model.py:7-60), B=1, T=3, 200x200
  * ``Decoder`` with the instance / centerness / flow heads ON (stp3/models/decoder.py:8-140), B=1
  * ``Encoder`` (both heads + restated trunk)   (stp3/models/encoder.py:57-97), 4 images
  * the WHOLE training step of BASELINE configs[2]: the reference's ``TrainingModule.shared_step``
    (stp3/trainer.py:101-172; imported unmodified, ``pytorch_lightning`` / ``fvcore`` stubbed) on a
    synthetic B=2 batch with ``LIFT.GT_DEPTH``, ``INSTANCE_SEG`` and ``INSTANCE_FLOW`` on: every entry
    of the loss dictionary, the summed loss, and a strided sample of every parameter's gradient.

Weights are name-derived (tests/helpers.fill_deterministic), inputs come from tests/helpers.det_tensor /
stp3_amd.synthetic, so the GPU box regenerates bit-identical weights and inputs without /root/reference.
The EfficientNet trunk and the ResNet-18 stages are the restated ones (parity UNPINNED for those two, see
make_golden_modules.py).
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

from oracle import ref_stubs  # noqa: E402
from stp3_amd import synthetic  # noqa: E402
from stp3_amd.config import CfgNode, perception_cfg  # noqa: E402
from stp3_amd.models.efficientnet import EfficientNet as OurEfficientNet  # noqa: E402
from stp3_amd.models.resnet import resnet18 as our_resnet18  # noqa: E402
from tests import helpers as H  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
C3 = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True}


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_trainer_stubs():
    """What stp3/trainer.py and stp3/metrics.py import on top of ref_stubs.install(): Lightning's module /
    metric base classes (plain nn.Module here: shared_step uses none of their machinery) and fvcore's CfgNode
    (the product's stand-in with the same behaviour, stp3_amd/config.py)."""

    class Metric(nn.Module):
        def __init__(self, compute_on_step=False, **kwargs):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            self.register_buffer(name, default, persistent=False)      # Lightning metric states are not checkpointed

    _mod('pytorch_lightning', LightningModule=nn.Module)
    _mod('pytorch_lightning.metrics')
    _mod('pytorch_lightning.metrics.metric', Metric=Metric)
    _mod('pytorch_lightning.metrics.functional')
    _mod('pytorch_lightning.metrics.functional.classification', stat_scores_multiple_classes=None)
    _mod('pytorch_lightning.metrics.functional.reduction', reduce=None)
    _mod('fvcore')
    _mod('fvcore.common')
    _mod('fvcore.common.config', CfgNode=CfgNode)


def make_deterministic_train(module):
    """train() with the stochastic layers neutralised (the same call is made on the product side)."""
    module.train()
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        gp = getattr(m, '_global_params', None)
        if gp is not None and hasattr(gp, 'drop_connect_rate'):
            gp.drop_connect_rate = 0.0
    return module


def grad_samples(module, prefix, out, n=256):
    for name, p in module.named_parameters():
        if p.grad is None:
            continue
        out[f'{prefix}/grad/{name}'] = H.sample(p.grad, n).numpy()
        out[f'{prefix}/gnorm/{name}'] = np.array([p.grad.double().norm().item()])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    ref_stubs.install(efficientnet_cls=OurEfficientNet, resnet18_fn=our_resnet18)
    install_trainer_stubs()
    from stp3.models.temporal_model import TemporalModel
    from stp3.models.decoder import Decoder
    from stp3.models.encoder import Encoder
    from stp3.trainer import TrainingModule

    out = {}
    t0 = time.time()

    # ---- TemporalModel, train mode, forward + backward --------------------------------------------------
    m = make_deterministic_train(H.fill_deterministic(TemporalModel(70, 3, input_shape=(200, 200), start_out_channels=64)))
    x = H.det_tensor((1, 3, 70, 200, 200), 21).requires_grad_(True)
    y = m(x)
    (y * H.det_tensor(tuple(y.shape), 22)).sum().backward()
    out['tm/out'] = H.sample(y).numpy()
    out['tm/dx'] = H.sample(x.grad).numpy()
    grad_samples(m, 'tm', out)
    print('TemporalModel done', round(time.time() - t0, 1), 's', flush=True)

    # ---- Decoder with every perception head on ----------------------------------------------------------
    gate = {'perceive_hdmap': True, 'predict_pedestrian': True, 'predict_instance': True,
            'predict_future_flow': True, 'planning': False}
    m = make_deterministic_train(H.fill_deterministic(Decoder(64, 2, 3, 2, gate)))
    x = H.det_tensor((1, 3, 64, 200, 200), 23).requires_grad_(True)
    o = m(x)
    heads = ('segmentation', 'pedestrian', 'hdmap', 'instance_center', 'instance_offset', 'instance_flow')
    loss = sum((o[k] * H.det_tensor(tuple(o[k].shape), 24 + i)).sum() for i, k in enumerate(heads))
    loss.backward()
    for k in heads:
        out[f'dec/{k}'] = H.sample(o[k]).numpy()
    out['dec/dx'] = H.sample(x.grad).numpy()
    grad_samples(m, 'dec', out)
    print('Decoder done', round(time.time() - t0, 1), 's', flush=True)

    # ---- Encoder (trunk + both heads), 4 images ---------------------------------------------------------
    cfg = perception_cfg()
    m = make_deterministic_train(H.fill_deterministic(Encoder(cfg.MODEL.ENCODER, D=48)))
    x = H.det_tensor((4, 3, 224, 480), 31).requires_grad_(True)
    f, d = m(x)
    ((f * H.det_tensor(tuple(f.shape), 32)).sum() + (d * H.det_tensor(tuple(d.shape), 33)).sum()).backward()
    out['enc/feat'], out['enc/depth'] = H.sample(f).numpy(), H.sample(d).numpy()
    out['enc/dx'] = H.sample(x.grad).numpy()
    grad_samples(m, 'enc', out)
    print('Encoder done', round(time.time() - t0, 1), 's', flush=True)

    # ---- the whole c3 training step through the reference's TrainingModule.shared_step --------------------
    cfg = perception_cfg(**C3)
    ref = TrainingModule(cfg.convert_to_dict())
    H.fill_deterministic(ref.model)
    make_deterministic_train(ref)
    batch = synthetic.make_batch(batch=2, seq=3, seed=5, gt_depth=True, instance=True)
    output, labels, loss = ref.shared_step(batch, True)
    total = sum(loss.values())
    total.backward()
    for k, v in loss.items():
        out[f'step/loss/{k}'] = np.array([v.item()], dtype=np.float64)
    out['step/loss_total'] = np.array([total.item()], dtype=np.float64)
    for k in ('segmentation', 'pedestrian', 'hdmap', 'instance_center', 'instance_offset', 'instance_flow',
              'depth_prediction'):
        out[f'step/out/{k}'] = H.sample(output[k]).numpy()
    for k in ('segmentation', 'pedestrian', 'instance', 'centerness', 'offset', 'flow', 'depths', 'hdmap'):
        out[f'step/label_sum/{k}'] = np.array([labels[k].double().sum().item()])
    grad_samples(ref.model, 'step', out)
    print('training step done', round(time.time() - t0, 1), 's  loss', total.item(), flush=True)

    np.savez_compressed(os.path.join(GOLDEN, 'train.npz'), **out)
    # the reference's parameter / buffer names and shapes (drop-in boundary B1): replayed by tests/test_state_dict_keys.py
    keys = {k: list(v.shape) for k, v in ref.state_dict().items()}
    json.dump({'TrainingModule_c3': keys}, open(os.path.join(GOLDEN, 'state_dict_keys.json'), 'w'), indent=0,
              sort_keys=True)
    man_path = os.path.join(GOLDEN, 'MANIFEST.json')
    man = json.load(open(man_path))
    man['train'] = {'file': 'train.npz', 'generator': 'oracle/make_golden_train.py',
                    'what': 'reference classes, float32 CPU, train() mode, Dropout p=0, drop-connect 0: '
                            'TemporalModel / Decoder (all heads) / Encoder forward+backward, and the reference '
                            'TrainingModule.shared_step at B=2 with the c3 overrides (loss dict + gradient samples)',
                    'loss_total': out['step/loss_total'].item(), 'entries': len(out)}
    man['state_dict_keys'] = {'file': 'state_dict_keys.json', 'count': len(keys)}
    json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)
    print(len(out), 'arrays;', len(keys), 'state-dict keys')


if __name__ == '__main__':
    main()
