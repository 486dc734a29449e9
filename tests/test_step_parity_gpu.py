"""GPU: the WHOLE benchmarked training step (BASELINE configs[2]) in float32 against the reference, block by block --
link 3 of the parity argument (link 1: tests/test_step_truth_cpu.py, the model mathematics in float64 against the
reference in float64, noise-free; link 2: tests/test_train_parity_gpu.py, every kernel block by block).

Fixtures: tests/golden/step_<variant>.npz, written by oracle/make_golden_step.py from the reference's own
``TrainingModule.shared_step`` (stp3/trainer.py:101-172) on the CPU -- loss dictionary, head outputs, a gradient
fingerprint of every parameter and, for every block (22 MBConv, 6 BasicBlock, 5 up-sampling, 2 TemporalBlock,
3 DeepLabHead, 6 decoder heads), fingerprints of the block's input, output and the gradient arriving at its output.
The same taps (tests/helpers.BlockTaps) are put on the product's step here.

What "equal" can mean for this step.  It is ill-conditioned in float32 (~130 train-mode BatchNorms and as many ReLUs): the
REFERENCE's float32 run (``b2k0``) differs from the REFERENCE's float64 evaluation (``b2k0d``, the truth) by ~4e-3 in the
decoder outputs and by 8..11 % in the temporal / encoder / trunk gradients, with the top-k selection of the losses
switched OFF.  That number is the reference's own rounding noise, not a property of any implementation; it is computed
here from the two fixtures, tap by tap and parameter group by parameter group, and it IS the tolerance:

  * ``test_step_b2_within_reference_noise``: the product's float32 step is CLOSER to the truth than the reference's
    float32 step is (<= 0.6 x the reference's distance, + 1e-4), for every loss entry, head output, block output, block
    gradient and parameter-gradient group -- measured 0.12 .. 0.18 x (profiles/r04a_parity_step.json: gradients 0.015 /
    0.012 / 0.010 / 0.0009 for trunk / encoder heads / temporal / decoder where the reference has 0.109 / 0.106 / 0.081 /
    0.009); and within 1.5 x of the reference's float32 values themselves (a run that sits at the truth is one
    reference-noise away from the reference: measured 1.03 .. 1.06 x).
  * ``b4k0`` / ``b4k1`` (the bench's batch size; ``b4k1`` IS configs[2], top-k on): float32 reference fixtures only (a
    float64 reference step at B=4 does not fit this container): losses to 2e-4, everything else within 3 x the B=2
    noise profile, the gradient groups within 1.5 x; ``b4k1`` twice: on the reference's OWN selection of the k hardest
    pixels (stored in the fixture) at the same bounds as ``b4k0`` -- measured temporal gradients 0.081 -- and with the
    product selecting for itself, where the temporal-model gradients additionally carry the pixels the two runs rank
    differently around the k-th largest loss (measured 0.25, bounded at 0.4: the difference between the two tests IS
    that re-selection).

Since round 4 every convolution of these float32 runs goes through the hand-written MFMA kernels (three-term bf16 split,
ops.conv2d_f32) instead of a vendor float32 convolution.  That settled a question of round 3: the temporal gradients'
distance from the truth had moved 0.027 -> 0.065 between two commits that changed no float32 arithmetic of the model.
scripts/step_noise_probe.py (profiles/r04a_step_noise_probe.json) perturbs the input images by one float32 ulp and
repeats the step: the distance moves between 0.009 and 0.038 from draw to draw (a factor of 4: the ReLU masks of ~130
BatchNorm + ReLU layers decide it), and it is the same to four digits with the torch statements of the losses in place of
the loss kernels (the change between those two commits).  The old numbers were two draws of that noise around the vendor
convolution's own, larger error; with the split-precision convolutions the run sits at 0.010 .. 0.038.
"""
import json
import os

import numpy as np
import pytest
import torch

from stp3_amd import synthetic
from stp3_amd.config import perception_cfg
from tests import helpers as H
from tests.test_train_parity_gpu import make_deterministic_train

pytestmark = pytest.mark.gpu
C3 = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True}
NO_TOPK = {'SEMANTIC_SEG.VEHICLE.USE_TOP_K': False, 'SEMANTIC_SEG.PEDESTRIAN.USE_TOP_K': False,
           'SEMANTIC_SEG.HDMAP.USE_TOP_K': [False, False]}
OUTPUTS = ('segmentation', 'pedestrian', 'hdmap', 'instance_center', 'instance_offset', 'instance_flow', 'depth_prediction')
GROUPS = [('encoder.backbone', 'trunk'), ('encoder', 'encoder_heads'), ('temporal_model', 'temporal'),
          ('decoder', 'decoder')]
NOISE_FACTOR_TRUTH, NOISE_FACTOR_REF, FLOOR = 0.6, 1.5, 1e-4
REPORT = {}
DEVICE = os.environ.get('STP3_PARITY_DEVICE', 'cuda')          # 'cpu': the product's plain-torch path (exploration only)
PERTURB = None              # (seed, relative size): scripts/step_noise_probe.py perturbs the images by that much


def rel(a, ref):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    r = torch.as_tensor(np.asarray(ref)).double().flatten()
    return ((a - r).norm() / r.norm().clamp_min(1e-30)).item()


# BASELINE configs[4] geometry on the one-camera rig of the ``c5...`` fixtures (oracle/make_golden_step.py: C5_GEOMETRY)
C5_GEOMETRY = {'IMAGE.FINAL_DIM': (896, 1600), 'LIFT.X_BOUND': [-50.0, 50.0, 0.25], 'LIFT.Y_BOUND': [-50.0, 50.0, 0.25],
               'LIFT.D_BOUND': [2.0, 66.0, 1.0]}
C5_BATCH = dict(n_cams=1, final_dim=(896, 1600), bev=(400, 400))


# BASELINE configs[0]: one frame, identity temporal model (the ``t1...`` fixtures of oracle/make_golden_step.py)
T1 = {'TIME_RECEPTIVE_FIELD': 1, 'MODEL.TEMPORAL_MODEL.NAME': 'identity'}


def run_product_step(variant):
    from stp3_amd.trainer import TrainingModule
    t1 = variant.startswith('t1')
    if t1:
        variant = variant[2:]
    c5 = variant.startswith('c5')
    body = variant[2:] if c5 else variant
    batch_size, topk = int(body[1:body.index('k')]), body.endswith('k1')
    over = dict(C3)
    if not topk:
        over.update(NO_TOPK)
    if c5:
        over.update(C5_GEOMETRY)
    if t1:
        over.update(T1)
    tm = TrainingModule(perception_cfg(**over).convert_to_dict())
    H.fill_deterministic(tm.model)
    make_deterministic_train(tm)
    tm = tm.to(DEVICE)
    taps = H.BlockTaps(tm.model)
    batch = synthetic.make_batch(batch=batch_size, seq=1 if t1 else 3, seed=5, gt_depth=True, instance=True, **(C5_BATCH if c5 else {}))
    if PERTURB is not None:
        noise = torch.randn(batch['image'].shape, generator=torch.Generator().manual_seed(PERTURB[0]))
        batch['image'] = batch['image'] * (1.0 + PERTURB[1] * noise)
    batch = {k: (v.to(DEVICE) if torch.is_tensor(v) and k not in ('intrinsics', 'extrinsics', 'future_egomotion') else v)
             for k, v in batch.items()}
    output, labels, loss = tm.shared_step(batch, True)
    for k in H.DECODER_HEADS:
        output[k].retain_grad()
    total = sum(loss.values())
    total.backward()
    fp = taps.collect()
    # the decoder's heads run through layers.fused.run_fused (no module call to hook): their taps are the head outputs
    # (the reference applies them to the frame-folded tensor; hdmap to the present frame only, decoder.py:122)
    for k, attr in H.DECODER_HEADS.items():
        o, g = output[k], output[k].grad
        if k != 'hdmap':
            o, g = o.flatten(0, 1), g.flatten(0, 1)
        fp[f'decoder.{attr}/out'], fp[f'decoder.{attr}/out_norm'] = H.fingerprint(o)
        fp[f'decoder.{attr}/gout'], fp[f'decoder.{attr}/gout_norm'] = H.fingerprint(g)
    return tm, output, labels, loss, total, fp


def measure(variant, fixture=None):
    """Distances of the product's float32 step from a fixture (default: the variant's own)."""
    g = H.load(f'step_{fixture or variant}.npz')
    tm, output, labels, loss, total, fp = run_product_step(variant)
    m = {'loss_total': abs(total.item() - g['loss_total'].item()) / abs(g['loss_total'].item())}
    for k, v in loss.items():
        ref = g[f'loss/{k}'].item()
        m[f'loss/{k}'] = abs(v.item() - ref) / max(abs(ref), 1e-3)
    assert {k[5:] for k in g.files if k.startswith('loss/')} == set(loss)
    for k in OUTPUTS:
        m[f'out/{k}'] = rel(H.sample(output[k], 256).cpu(), g[f'out/{k}'])
    got = {n: H.sample(p.grad, 256).double().cpu() for n, p in tm.model.named_parameters() if p.grad is not None}
    missing = [k[7:] for k in g.files if k.startswith('p/grad/') and k[7:] not in got]
    assert not missing, missing[:5]
    m.update({f'grad/{k}': v for k, v in group_errors(got, g).items()})
    blocks = sorted({k.rsplit('/', 1)[0] for k in g.files if k.endswith('/out')})
    assert len(blocks) >= (41 if variant.startswith('t1') else 44), len(blocks)
    for b in blocks:
        assert f'{b}/out' in fp, f'no tap on {b}'
        m[f'tap_out/{b}'] = rel(fp[f'{b}/out'], g[f'{b}/out'])
        if f'{b}/gout' in g.files:
            m[f'tap_gout/{b}'] = rel(fp[f'{b}/gout'], g[f'{b}/gout'])
    return m, (got, fp, output, loss, total)


def group_errors(grads, g):
    """{group: relative L2 over the gradient fingerprints of the group's parameters} of ``grads`` (name -> sample) vs g."""
    acc = {}
    for name, a in grads.items():
        key = f'p/grad/{name}'
        if key not in g.files:
            continue
        grp = next((v for k, v in GROUPS if name.startswith(k)), 'other')
        e = acc.setdefault(grp, [[], []])
        e[0].append(torch.as_tensor(np.asarray(a)).double().flatten())
        e[1].append(torch.from_numpy(g[key]).double().flatten())
    return {k: ((torch.cat(a) - torch.cat(r)).norm() / torch.cat(r).norm()).item() for k, (a, r) in acc.items()}


def reference_noise(f32='step_b2k0.npz', truth='step_b2k0d.npz'):
    """The same distances for the REFERENCE's float32 step (step_b2k0) from the truth (step_b2k0d): its rounding noise."""
    a, d = H.load(f32), H.load(truth)
    n = {'loss_total': abs(a['loss_total'].item() - d['loss_total'].item()) / abs(d['loss_total'].item())}
    for k in a.files:
        if k.startswith('loss/'):
            n[k] = abs(a[k].item() - d[k].item()) / max(abs(d[k].item()), 1e-3)
        elif k.startswith('out/'):
            n[k] = rel(a[k], d[k])
        elif k.endswith('/out') or k.endswith('/gout'):
            b, kind = k.rsplit('/', 1)
            n[f'tap_{kind}/{b}'] = rel(a[k], d[k])
    n.update({f'grad/{k}': v for k, v in group_errors({k[7:]: a[k] for k in a.files if k.startswith('p/grad/')}, d).items()})
    return n


def report(tag, m, extra=None):
    REPORT[tag] = dict(m, **(extra or {}))
    path = os.environ.get('STP3_PARITY_REPORT_STEP')
    if path:
        json.dump(REPORT, open(path, 'w'), indent=1, sort_keys=True)
    grads = ', '.join(f'{k[5:]}={v:.2e}' for k, v in m.items() if k.startswith('grad/'))
    print(f'[step parity] {tag}: loss {m["loss_total"]:.2e}, grads {grads}, worst tap out '
          f'{max(v for k, v in m.items() if k.startswith("tap_out/")):.2e}, worst tap gout '
          f'{max(v for k, v in m.items() if k.startswith("tap_gout/")):.2e}')


def test_step_b2_within_reference_noise():
    noise = reference_noise()
    m_truth, _ = measure('b2k0', fixture='b2k0d')
    report('b2k0_vs_truth', m_truth, {'reference_noise': noise})
    bad = {k: (v, noise[k]) for k, v in m_truth.items() if v > NOISE_FACTOR_TRUTH * noise.get(k, 0.0) + FLOOR}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])


def test_step_b2_against_the_float32_reference():
    noise = reference_noise()
    m_ref, _ = measure('b2k0')
    report('b2k0_vs_reference_f32', m_ref)
    bad = {k: (v, noise[k]) for k, v in m_ref.items() if v > NOISE_FACTOR_REF * noise.get(k, 0.0) + FLOOR}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])


def test_step_t1_identity_temporal_model_configs0():
    """BASELINE configs[0]: T = 1, the IDENTITY temporal model (stp3/models/stp3.py:34-35; temporal_model.py:63-71) -- the
    whole float32 step on the kernels at one frame per sample against the reference's float64 run of that configuration
    (step_t1b2k0d.npz), bounded tap by tap by the reference's OWN float32 noise at that configuration (step_t1b2k0.npz vs
    the truth), and against the float32 fixture at that noise."""
    noise = reference_noise('step_t1b2k0.npz', 'step_t1b2k0d.npz')
    m_truth, _ = measure('t1b2k0', fixture='t1b2k0d')
    report('t1b2k0_vs_truth', m_truth, {'reference_noise': noise})

    # forward quantities (losses, head outputs, block outputs): the bound of the T = 3 test.  Gradient quantities: the noise
    # profile here is ONE draw of the reference on 12 images, and one draw of a gradient distance scatters by 2-4x around its
    # typical size (scripts/step_noise_probe.py, profiles/r04a_step_noise_probe.json): 3x that draw
    def factor(key, forward):
        return forward if key.startswith(('loss', 'out/', 'tap_out/')) else 3.0
    bad = {k: (v, noise[k]) for k, v in m_truth.items() if v > factor(k, NOISE_FACTOR_TRUTH) * noise.get(k, 0.0) + FLOOR}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])
    m_ref, _ = measure('t1b2k0')
    report('t1b2k0_vs_reference_f32', m_ref)
    bad = {k: (v, noise[k]) for k, v in m_ref.items() if v > factor(k, NOISE_FACTOR_REF) * noise.get(k, 0.0) + FLOOR}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])


def _check_b4(variant, temporal_bound, tag=None, own_selection=True):
    noise = reference_noise()                                  # the B=2 profile
    m, _ = measure(variant)
    report(tag or f'{variant}_vs_reference_f32', m)
    assert max(v for k, v in m.items() if k.startswith('loss')) <= 2e-4, {k: v for k, v in m.items() if k.startswith('loss')}
    worst = lambda prefix: max(v for k, v in noise.items() if k.startswith(prefix))
    bounds = {'out/': worst('out/'), 'tap_out/': worst('tap_out/'), 'tap_gout/': worst('tap_gout/')}
    for prefix, nz in bounds.items():
        if variant.endswith('k1') and own_selection and prefix == 'tap_gout/':
            continue                                           # top-k re-selection: bounded through the gradient groups below
        over = {k: v for k, v in m.items() if k.startswith(prefix) and v > 3.0 * nz + FLOOR}
        assert not over, (prefix, nz, dict(sorted(over.items(), key=lambda kv: -kv[1])[:5]))
    for grp in ('decoder', 'encoder_heads', 'trunk'):
        assert m[f'grad/{grp}'] <= NOISE_FACTOR_REF * noise[f'grad/{grp}'] + FLOOR, (grp, m[f'grad/{grp}'], noise[f'grad/{grp}'])
    assert m['grad/temporal'] <= temporal_bound, m['grad/temporal']


def test_step_b4_smooth():
    """The bench's batch size with the top-k selection off."""
    _check_b4('b4k0', temporal_bound=NOISE_FACTOR_REF * reference_noise()['grad/temporal'] + FLOOR)


def forward_noise(f32_name, truth_name):
    """Distances of the reference's float32 FORWARD (losses, head outputs, block outputs) from its float64 forward."""
    a, d = H.load(f32_name), H.load(truth_name)
    n = {'loss_total': abs(a['loss_total'].item() - d['loss_total'].item()) / abs(d['loss_total'].item())}
    for k in d.files:
        if k.startswith('loss/'):
            n[k] = abs(a[k].item() - d[k].item()) / max(abs(d[k].item()), 1e-3)
        elif k.startswith('out/'):
            n[k] = rel(a[k], d[k])
        elif k.endswith('/out'):
            n[f'tap_out/{k[:-4]}'] = rel(a[k], d[k])
    return n


def test_step_c5_geometry():
    """BASELINE configs[4] GEOMETRY through the whole float32 step -- 896 x 1600 images, 112-row image columns, 64 depth
    bins, 400 x 400 BEV cells of 0.25 m -- on the rig the reference's float32 run fits into this container's memory (one
    camera, T = 3, one sample; oracle/make_golden_step.py: C5_GEOMETRY), every block tapped.  The lift runs on the general
    column kernels here (fH = 112 > 32), the convolutions on the MFMA kernels at 448 x 800 .. 56 x 100 and 400 x 400.

    FORWARD, against the reference's FLOAT64 forward of the same step (``step_c5b1k0D.npz``: forward only -- its float64
    backward does not fit 62 GB): every loss entry, head output and block output within 0.3 x the distance of the
    reference's own float32 forward from that truth (+ 1e-4) -- measured (profiles/r04h_parity_step_c5.json): the decoder
    blocks 1e-4 .. 2e-4 where the reference's float32 run is 1e-2 .. 3e-2 off, i.e. 100 x closer.  The reference's noise
    is large on this rig: a one-camera BEV is 5/6 empty, the BEV-stage
    BatchNorms normalise populations of mostly identical values, the pooled-descriptor BatchNorms B * T = 3 vectors, and
    the reference's float32 prefix-sum pooling of 1.4 M points per frame is lossy.
    BACKWARD: there is no truth for this geometry; the gradients are recorded against the reference's float32 run (whose
    forward already sits 1e-2 off the truth in the decoder: its ReLU masks differ) and bounded loosely -- a sign error or
    a missing term would exceed it -- together with the product's own sensitivity to a ONE-ulp perturbation of the
    images, which is recorded beside them."""
    global PERTURB
    noise = forward_noise('step_c5b1k0.npz', 'step_c5b1k0D.npz')
    # one product run, measured against both fixtures
    g32, g64 = H.load('step_c5b1k0.npz'), H.load('step_c5b1k0D.npz')
    tm, output, labels, loss, total, fp = run_product_step('c5b1k0')
    m = {'loss_total': abs(total.item() - g64['loss_total'].item()) / abs(g64['loss_total'].item())}
    for k, v in loss.items():
        m[f'loss/{k}'] = abs(v.item() - g64[f'loss/{k}'].item()) / max(abs(g64[f'loss/{k}'].item()), 1e-3)
    for k in OUTPUTS:
        m[f'out/{k}'] = rel(H.sample(output[k], 256).cpu(), g64[f'out/{k}'])
    for key in g64.files:
        if key.endswith('/out'):
            m[f'tap_out/{key[:-4]}'] = rel(fp[key], g64[key])
    got = {n: H.sample(p.grad, 256).double().cpu() for n, p in tm.model.named_parameters() if p.grad is not None}
    grads_vs_ref = {f'grad/{k}': v for k, v in group_errors(got, g32).items()}
    gout_vs_ref = {f'tap_gout/{k[:-5]}': rel(fp[k], g32[k]) for k in g32.files if k.endswith('/gout')}
    del tm, output, labels, loss, total
    PERTURB = (100, 1e-7)
    try:
        fp2 = run_product_step('c5b1k0')[5]
    finally:
        PERTURB = None
    self_noise = {k: rel(fp2[k[9:] + '/gout'], fp[k[9:] + '/gout']) for k in gout_vs_ref}
    report('c5b1k0_forward_vs_truth', dict(m, **{'tap_gout/none': 0.0}),
           {'reference_forward_noise': noise, 'gradients_vs_reference_f32': grads_vs_ref, 'gout_vs_reference_f32': gout_vs_ref,
            'gout_self_noise_one_ulp': self_noise})
    bad = {k: (v, noise.get(k, 0.0)) for k, v in m.items() if v > 0.3 * noise.get(k, 0.0) + FLOOR}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1][0])[:8])
    assert max(grads_vs_ref.values()) <= 1.5, grads_vs_ref
    assert all(torch.isfinite(torch.as_tensor(v)).all() for v in got.values())


def test_step_b4_configs2():
    """BASELINE configs[2] exactly (top-k on), the product choosing its own k hardest pixels: the losses agree to 2e-4;
    the gradients behind the heads additionally carry the pixels that the two float32 runs rank differently around the
    k-th largest loss (the next test removes exactly that and nothing else)."""
    _check_b4('b4k1', temporal_bound=0.4)


def test_step_b4_configs2_on_the_references_selection(monkeypatch):
    """BASELINE configs[2] with the top-k losses evaluated on THE REFERENCE'S selected pixels (the fixture stores, for
    each of its three sorting losses -- vehicle, pedestrian, first hd-map element: losses.py:76-81, :108-111 -- which
    k = 10 000 of the 40 000 pixels of every row its descending sort kept): configs[2] pinned at the level of the smooth
    ``b4k0`` case, gradient taps included.  The selection enters through the loss KERNEL's own interface: the pixels the
    reference did not select get the ignore label (zero loss, zero gradient: stp3_ce_topk_fwd), the mean runs over all
    pixels and is rescaled by P / k."""
    from stp3_amd import ops_loss
    g = H.load('step_b4k1.npz')
    n_masks = len([k for k in g.files if k.startswith('topk/') and k.endswith('/mask')])
    assert n_masks == 3, n_masks
    masks = [(torch.from_numpy(np.unpackbits(g[f'topk/{i}/mask'], axis=1).astype(bool)), int(g[f'topk/{i}/k'][0]))
             for i in range(n_masks)]
    real, used = ops_loss.ce_topk_mean, []

    def on_the_references_pixels(logits, labels, class_weights=None, row_scale=None, top_k=0, ignore_index=255):
        h, w = logits.shape[-2:]
        if not 0 < int(top_k) < h * w:
            return real(logits, labels, class_weights, row_scale, top_k, ignore_index)
        mask, k = masks[len(used)]
        used.append(k)
        assert k == int(top_k) and mask.numel() == labels.numel(), (k, top_k, mask.shape, labels.shape)
        assert int(mask.sum()) == k * mask.shape[0]
        keep = mask[:, :h * w].reshape(labels.shape).to(labels.device)
        forced = torch.where(keep, labels, torch.full_like(labels, ignore_index))
        return real(logits, forced, class_weights, row_scale, 0, ignore_index) * (float(h * w) / k)
    monkeypatch.setattr(ops_loss, 'ce_topk_mean', on_the_references_pixels)
    _check_b4('b4k1', temporal_bound=NOISE_FACTOR_REF * reference_noise()['grad/temporal'] + FLOOR,
              tag='b4k1_on_reference_selection_vs_reference_f32', own_selection=False)
    assert len(used) == 3, used
