"""TEST INFRASTRUCTURE (build container only) -- whole-step fixtures of BASELINE configs[2] from the REFERENCE, with a
tap on every block.

    python oracle/make_golden_step.py b2k0 b4k1 b4k0      -> tests/golden/step_<variant>.npz

Runs the reference's own ``TrainingModule.shared_step`` (stp3/trainer.py:101-172, imported unmodified through
oracle/ref_stubs.py) on the CPU in float32, train() mode, Dropout p = 0 / drop-connect 0 (make_golden_train.py), on the
synthetic c3 batch, and records for each variant

  * every entry of the loss dictionary and the total,
  * a 256-sample fingerprint + norm of every head output and of every parameter's gradient,
  * for every block (22 MBConv, 6 BasicBlock, UpsamplingConcat x2, UpsamplingAdd x3, TemporalBlock x2, DeepLabHead x3,
    the 6 decoder heads): 512-sample fingerprints + norms of its input, its output and the gradient arriving at its
    output (forward hooks + ``retain_grad`` -- tests/helpers.BlockTaps, the same code the GPU test runs on the product).

Variants: ``[t1][c5]b<B>k<0|1>[d|D]`` (``D``: float64, FORWARD only) = (``t1``: BASELINE configs[0], one frame and the identity
temporal model; ``c5``: BASELINE configs[4] geometry on a one-camera rig, see C5_GEOMETRY) batch size B, top-k selection of the segmentation losses off / on
(SEMANTIC_SEG.*.USE_TOP_K, stp3/losses.py:43-76), ``d`` = the reference evaluated in FLOAT64.  ``b4k1`` IS configs[2],
reference unmodified, float32.

Why ``d`` exists.  The whole step is an ill-conditioned function in float32: ~130 train-mode BatchNorm layers and as
many ReLUs.  Measured on the reference ITSELF (``b2k0`` vs ``b2k0d``, same code, same weights, same batch): float32
rounding alone moves the reference's decoder outputs by ~3e-3 and, ReLU masks being discontinuous (a relative
perturbation d flips a fraction ~d of the units, each an O(1) change of its gradient: relative L2 ~ sqrt(d)), its
gradients by 5..15 % -- entering at the decoder's ReLUs and then flat through the 22 trunk blocks.  No float32
implementation, the reference on another thread count included, reproduces the reference's float32 gradients better
than that.  The ``d`` variants are the noise-free limit of the reference: its own modules in float64 -- image encoder,
temporal model, decoder, losses; the voxel pool's prefix sum runs on the float64 point matrix (the reference's own
``VoxelsSumming``) while the GEOMETRY stays float32 so that every point lands in the voxel the float32 reference puts
it in (voxel ids are bit-exact by contract).  tests/test_step_parity_gpu.py requires the product's float32 step to be
as close to that truth as the reference's own float32 step is (x a small factor), tap by tap and parameter group by
parameter group; the tight kernel-level pins are the single-module and block-level tests, where the conditioning is
benign.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

from oracle import ref_stubs  # noqa: E402
from oracle.make_golden_train import C3, grad_samples, install_trainer_stubs, make_deterministic_train  # noqa: E402
from stp3_amd import synthetic  # noqa: E402
from stp3_amd.config import perception_cfg  # noqa: E402
from stp3_amd.models.efficientnet import EfficientNet as OurEfficientNet  # noqa: E402
from stp3_amd.models.resnet import resnet18 as our_resnet18  # noqa: E402
from tests import helpers as H  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
NO_TOPK = {'SEMANTIC_SEG.VEHICLE.USE_TOP_K': False, 'SEMANTIC_SEG.PEDESTRIAN.USE_TOP_K': False,
           'SEMANTIC_SEG.HDMAP.USE_TOP_K': [False, False]}


def to_float64(ref):
    """The reference's learned modules in float64, its geometry (frustum, BEV grid, poses) left in float32: the lift
    puts every point into the same voxel as the float32 reference.  The reference scatters the pooled sums into float32
    tensors (stp3.py:280-292), so its own ``VoxelsSumming`` (stp3/utils/geometry.py:299-330) is called on the float64
    point matrix and its result rounded to float32 ONCE (relative 6e-8 -- not the ~4e-4 of a float32 prefix sum over
    450 000 points); the BEV tensor is cast up again in front of the temporal model.  Returns the undo function."""
    import stp3.models.stp3 as ref_stp3
    from stp3.utils.geometry import VoxelsSumming as RefVS

    class VoxelsSummingF64(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, geometry, ranks):
            ctx.x_dtype = x.dtype
            out, geo = RefVS.forward(ctx, x.double(), geometry, ranks)
            ctx.mark_non_differentiable(geo)
            return out.float(), geo

        @staticmethod
        def backward(ctx, grad_x, grad_geometry):
            return RefVS.backward(ctx, grad_x, grad_geometry)[0].to(ctx.x_dtype), None, None

    ref_stp3.VoxelsSumming = VoxelsSummingF64
    m = ref.model
    for sub in (m.encoder, m.temporal_model, m.decoder):
        sub.double()
    m.temporal_model.register_forward_pre_hook(lambda mod, args: tuple(a.double() if torch.is_tensor(a) else a for a in args))
    for loss_fn in ref.losses_fn.values():                       # class weights are plain tensor attributes (losses.py:46,88)
        w = getattr(loss_fn, 'class_weights', None)
        if torch.is_tensor(w):
            loss_fn.class_weights = w.double()
        # float32 regression targets (centerness / offset / flow labels) meet float64 predictions: l1 / mse backward
        # want one dtype
        loss_fn.register_forward_pre_hook(lambda mod, args: tuple(
            a.double() if (torch.is_tensor(a) and a.is_floating_point()) else a for a in args))
    return lambda: setattr(ref_stp3, 'VoxelsSumming', RefVS)


def topk_ratios(over):
    """The top-k ratios in the order the reference's losses sort (trainer.py:122-150: segmentation, pedestrian, then the
    hdmap elements)."""
    cfg = perception_cfg(**over)
    seg = cfg.SEMANTIC_SEG
    out = []
    if seg.VEHICLE.USE_TOP_K:
        out.append(seg.VEHICLE.TOP_K_RATIO)
    if seg.PEDESTRIAN.USE_TOP_K:
        out.append(seg.PEDESTRIAN.TOP_K_RATIO)
    out += [r for r, use in zip(seg.HDMAP.TOP_K_RATIO, seg.HDMAP.USE_TOP_K) if use]
    return out


# BASELINE configs[4] geometry: 896 x 1600 images (fH x fW = 112 x 200), D = 64 depth bins, 400 x 400 BEV cells of 0.25 m --
# on a rig this container can hold in float32: ONE camera, T = 3 frames, one sample (3 images of 1.4 MP; the six-camera,
# five-frame sample of the configuration needs > 100 GB of float32 activations on the CPU).  Every operator of the step
# runs at the configuration's sizes: the trunk at 448 x 800 .. 56 x 100, the lift with 112-row columns and 64 bins into
# 160 000 cells, the temporal model and the decoder on 400 x 400 maps.
C5_GEOMETRY = {'IMAGE.FINAL_DIM': (896, 1600), 'LIFT.X_BOUND': [-50.0, 50.0, 0.25], 'LIFT.Y_BOUND': [-50.0, 50.0, 0.25],
               'LIFT.D_BOUND': [2.0, 66.0, 1.0]}
C5_BATCH = dict(n_cams=1, final_dim=(896, 1600), bev=(400, 400))


# BASELINE configs[0]: T = 1, i.e. the IDENTITY temporal model (stp3/models/stp3.py:34-35, 56-57: with a receptive field of one
# frame the configuration must name 'identity'; TemporalModelIdentity hands the 64 + 6 ego-motion channels of the present
# frame to the decoder) -- the path on which the 70-channel concatenation really exists.
T1 = {'TIME_RECEPTIVE_FIELD': 1, 'MODEL.TEMPORAL_MODEL.NAME': 'identity'}


def variant_cfg(variant):
    variant = variant.rstrip('dD')
    t1 = variant.startswith('t1')
    if t1:
        variant = variant[2:]
    c5 = variant.startswith('c5')
    body = variant[2:] if c5 else variant
    batch, topk = int(body[1:body.index('k')]), body.endswith('k1')
    over = dict(C3)
    if not topk:
        over.update(NO_TOPK)
    if c5:
        over.update(C5_GEOMETRY)
    if t1:
        over.update(T1)
    return batch, over


def run_variant(variant, TrainingModule):
    batch_size, over = variant_cfg(variant)
    forward_only = variant.endswith('D')                 # float64, forward pass only (no-grad: what fits at configs[4] sizes)
    f64 = variant.endswith('d') or forward_only
    t0 = time.time()
    ref = TrainingModule(perception_cfg(**over).convert_to_dict())
    H.fill_deterministic(ref.model)
    make_deterministic_train(ref)
    restore = to_float64(ref) if f64 else (lambda: None)
    heads = [f'decoder.{a}' for a in H.DECODER_HEADS.values()]
    taps = H.BlockTaps(ref.model, extra=heads, forward_only=forward_only)
    batch = synthetic.make_batch(batch=batch_size, seq=1 if variant.startswith('t1') else 3, seed=5, gt_depth=True, instance=True,
                                 **(C5_BATCH if 'c5' in variant[:4] else {}))
    if f64:
        batch['image'] = batch['image'].double()
    # the pixels the reference's top-k losses SELECT (losses.py:76-81, :108-111: a descending sort, the first k kept):
    # recorded from its own torch.sort calls, so that a test can hand the product the very same selection
    picked, real_sort = [], torch.sort

    def recording_sort(x, *a, **kw):
        res = real_sort(x, *a, **kw)
        if kw.get('descending') and x.dim() in (2, 3):
            picked.append(res[1].detach())
        return res
    torch.sort = recording_sort
    try:
        with torch.set_grad_enabled(not forward_only):
            output, labels, loss = ref.shared_step(batch, True)
    finally:
        torch.sort = real_sort
    total = sum(loss.values())
    if not forward_only:
        total.backward()
    out = dict(taps.collect())
    for i, idx in enumerate(picked):
        # the reference keeps idx[..., :k] with k = int(ratio * P): stored as one bit per pixel and row
        rows = idx.reshape(-1, idx.shape[-1])
        ratio = topk_ratios(over)[i]
        k = int(ratio * rows.shape[1])
        mask = torch.zeros(rows.shape, dtype=torch.bool)
        mask.scatter_(1, rows[:, :k], True)
        out[f'topk/{i}/mask'] = np.packbits(mask.numpy(), axis=1)
        out[f'topk/{i}/k'] = np.array([k], dtype=np.int64)
    for k, v in loss.items():
        out[f'loss/{k}'] = np.array([v.item()], dtype=np.float64)
    out['loss_total'] = np.array([total.item()], dtype=np.float64)
    for k in ('segmentation', 'pedestrian', 'hdmap', 'instance_center', 'instance_offset', 'instance_flow',
              'depth_prediction'):
        out[f'out/{k}'] = H.sample(output[k], 256).numpy()
    for k in ('segmentation', 'pedestrian', 'instance', 'centerness', 'offset', 'flow', 'depths', 'hdmap'):
        out[f'label_sum/{k}'] = np.array([labels[k].double().sum().item()])
    if not forward_only:
        grad_samples(ref.model, 'p', out)
    restore()
    out = {k: (v.astype(np.float32) if v.dtype == np.float64 and v.size > 1 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(GOLDEN, f'step_{variant}.npz'), **out)
    print(f'{variant}: loss {total.item():.6f}, {len(out)} arrays, {len(taps.names)} blocks, '
          f'{time.time() - t0:.0f} s', flush=True)
    return {'file': f'step_{variant}.npz', 'generator': 'oracle/make_golden_step.py', 'batch': batch_size,
            'top_k': variant.rstrip('d').endswith('k1'), 'float64': f64, 'loss_total': total.item(), 'entries': len(out),
            'what': 'reference TrainingModule.shared_step, float32 CPU, train() mode, c3 overrides: loss dict, head '
                    'outputs, gradient fingerprints of every parameter, in / out / grad-out fingerprints of every block'}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    ref_stubs.install(efficientnet_cls=OurEfficientNet, resnet18_fn=our_resnet18)
    install_trainer_stubs()
    from stp3.trainer import TrainingModule
    man_path = os.path.join(GOLDEN, 'MANIFEST.json')
    for variant in sys.argv[1:] or ['b2k0', 'b4k1', 'b4k0']:
        entry = run_variant(variant, TrainingModule)
        man = json.load(open(man_path))
        man[f'step_{variant}'] = entry
        json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
