"""CPU: host-side pieces of the path against the fixtures generated from the reference's own classes
(tests/golden/modules.npz, oracle/make_golden_modules.py) -- the dense modules through the plain-torch
statement of the fused operators, the losses, the label warp and the IoU metric.  These are the same
fixtures tests/test_modules_gpu.py replays on the MI355X through the HIP kernels.

Tolerance: float32 CPU vs float32 CPU of the reference (same torch build in the container that generated the
fixtures): rtol 2e-3 / atol 2e-4 as on the GPU (other BLAS / thread counts may reorder sums); losses rtol 1e-5.
"""
import pytest
import torch

from stp3_amd import synthetic
from tests import helpers as H

G = H.load('modules.npz')


def close(actual, key, rtol=2e-3, atol=2e-4):
    torch.testing.assert_close(H.sample(actual).float(), torch.from_numpy(G[key]), rtol=rtol, atol=atol)


def prep(m):
    return H.fill_deterministic(m).eval()


@torch.no_grad()
def test_heads_upsampling_and_temporal_block_match_the_reference_modules():
    from stp3_amd.layers.convolutions import DeepLabHead, UpsamplingAdd, UpsamplingConcat
    from stp3_amd.layers.temporal import TemporalBlock
    close(prep(DeepLabHead(160, 160, hidden_channel=64))(H.det_tensor((2, 160, 14, 30), 1)), 'deeplab_enc')
    close(prep(UpsamplingConcat(216, 64))(H.det_tensor((2, 160, 14, 30), 3), H.det_tensor((2, 56, 28, 60), 4)), 'upconcat')
    close(prep(UpsamplingAdd(256, 128))(H.det_tensor((2, 256, 25, 25), 5), H.det_tensor((2, 128, 50, 50), 6)), 'upadd')
    m = prep(TemporalBlock(70, 64, use_pyramid_pooling=True, pool_sizes=[(2, 40, 40)]))
    close(m(H.det_tensor((2, 70, 3, 40, 40), 7)), 'tblock')


def test_losses_and_label_warp_match_the_reference():
    from stp3_amd import geometry as geo
    from stp3_amd import losses as L
    pred = H.det_tensor((2, 3, 2, 200, 200), 11, 3.0)
    seg, ped, hd = synthetic.make_labels(2, 3, seed=4)
    l1 = L.SegmentationLoss(torch.Tensor([1.0, 2.0]), use_top_k=True, top_k_ratio=0.25, future_discount=0.95)(pred, seg, 3)
    l2 = L.HDmapLoss(torch.Tensor([[1.0, 5.0], [1.0, 1.0]]), [1, 1], [True, False], [0.25, 0.25])(
        H.det_tensor((2, 4, 200, 200), 12, 3.0), hd[:, 2])
    l3 = L.DepthLoss()(H.det_tensor((1, 2, 2, 48, 28, 60), 13, 3.0), (H.det_tensor((1, 2, 2, 28, 60), 14).abs() * 47).long())
    tgt = H.det_tensor((2, 3, 2, 50, 50), 15)
    tgt[:, :, :, :10] = 255
    l4 = L.SpatialRegressionLoss(norm=1, future_discount=0.95)(H.det_tensor((2, 3, 2, 50, 50), 16), tgt, 2)
    got = torch.stack([l1, l2, l3, l4]).double()
    torch.testing.assert_close(got, torch.from_numpy(G['losses']), rtol=1e-5, atol=1e-6)
    ego = synthetic.make_rig(2, 3, seed=6)[2]
    wp = geo.cumulative_warp_features(seg.float(), ego, 'nearest', (50.0, 50.0))
    wr = geo.cumulative_warp_features_reverse(seg.float(), ego, 'nearest', (50.0, 50.0))
    torch.testing.assert_close(wp.sum(dim=(-1, -2, -3)), torch.from_numpy(G['warp_past_sum']))
    torch.testing.assert_close(wr.sum(dim=(-1, -2, -3)), torch.from_numpy(G['warp_rev_sum']))
    assert torch.equal(H.sample(wp), torch.from_numpy(G['warp_past_sample']))


def test_iou_metric_counts():
    from stp3_amd.metrics import IntersectionOverUnion
    g = torch.Generator().manual_seed(2)
    pred = (torch.rand(2, 1, 1, 50, 50, generator=g) > 0.6).long()
    tgt = (torch.rand(2, 1, 1, 50, 50, generator=g) > 0.5).long()
    m = IntersectionOverUnion(2)
    m(pred, tgt)
    tp = int(((pred == 1) & (tgt == 1)).sum())
    fp = int(((pred == 1) & (tgt == 0)).sum())
    fn = int(((pred == 0) & (tgt == 1)).sum())
    assert abs(float(m.compute()[1]) - tp / (tp + fp + fn)) < 1e-6          # stp3/metrics.py:37-65
    # an independent implementation of the same quantity (scikit-learn's Jaccard index), accumulated over two updates
    sk = pytest.importorskip('sklearn.metrics')
    m.reset()
    m(pred[:1], tgt[:1])
    m(pred[1:], tgt[1:])
    want = sk.jaccard_score(tgt.reshape(-1).numpy(), pred.reshape(-1).numpy(), average=None)
    assert abs(float(m.compute()[0]) - want[0]) < 1e-6 and abs(float(m.compute()[1]) - want[1]) < 1e-6


def test_staged_sum_equals_sum():
    from stp3_amd.utils import staged_mean, staged_sum
    g = torch.Generator().manual_seed(0)
    for n in (1, 4095, 4096, 8193, 1_000_003):
        x = torch.randn(n, generator=g, dtype=torch.float64)
        torch.testing.assert_close(staged_sum(x), x.sum(), rtol=1e-12, atol=1e-9)
        torch.testing.assert_close(staged_mean(x), x.mean(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('train', [True, False])
def test_bn_act_reference_equals_torch_modules(train):
    import torch.nn as nn
    from stp3_amd.layers import fused
    torch.manual_seed(0)
    bn, ref = nn.BatchNorm2d(12), nn.BatchNorm2d(12)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_var.uniform_(0.5, 1.5)
    ref.load_state_dict(bn.state_dict())
    bn.train(train)
    ref.train(train)
    x = torch.randn(3, 12, 5, 7, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    res = torch.randn(3, 12, 5, 7)
    y = fused.bn_act(bn, x, fused.ACT_SWISH, res=res, res_mode=fused.RES_AFTER_ACT)
    y2 = torch.nn.functional.silu(ref(x2)) + res
    torch.testing.assert_close(y, y2, rtol=1e-5, atol=1e-5)
    y.sum().backward()
    y2.sum().backward()
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_var, ref.running_var)


def test_batched_label_warp_equals_per_label_warps():
    """STP3_LABEL_WARP=batched: pose chains once per batch + one grid_sample per frame over all label maps ==
    the reference-shaped per-label ``cumulative_warp_features`` calls (trainer.py:254-360), bit for bit."""
    import torch
    from stp3_amd import synthetic, trainer
    from stp3_amd.config import perception_cfg
    for extra in ({}, {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True}):
        cfg = perception_cfg(**{'IMAGE.FINAL_DIM': (64, 96), **extra})
        module = trainer.TrainingModule(cfg.convert_to_dict())
        full = bool(extra)
        batch = synthetic.make_batch(batch=3, seq=3, final_dim=(64, 96), seed=4, gt_depth=full, instance=full)
        batch['future_egomotion'][..., 5] *= 5.0            # a real turn: the warps move pixels
        old = module.prepare_future_labels(batch)
        new = module._prepare_future_labels_batched(batch)
        assert old.keys() == new.keys()
        for k in old:
            assert old[k].dtype == new[k].dtype and torch.equal(old[k], new[k]), k
        assert (old['segmentation'][:, 0] != batch['segmentation'][:, 0]).any()


def test_lazy_batch_counters_match_immediate_increments(monkeypatch):
    """STP3_LAZY_BN_COUNTER: num_batches_tracked counted on the host, applied by flush / before state_dict()."""
    import torch
    import torch.nn as nn
    from stp3_amd import ops
    from stp3_amd.layers import fused
    monkeypatch.setattr(ops, 'LAZY_COUNTERS', True)
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    monkeypatch.setattr(ops, 'bn_act', lambda x, *a, **k: x)              # the kernel path is not under test here
    a, b, c = nn.BatchNorm2d(4), nn.BatchNorm2d(4), nn.BatchNorm2d(4, momentum=None)
    x = torch.randn(2, 4, 3, 3)
    for _ in range(3):
        fused.bn_act(a, x)
    fused.bn_act(b, x)
    fused.bn_act(c, x)                                    # cumulative-average mode: immediate
    assert int(c.num_batches_tracked) == 1 and int(a.num_batches_tracked) == 0
    assert int(a.state_dict()['num_batches_tracked']) == 3          # the pre-hook flushed every pending counter
    assert int(b.num_batches_tracked) == 1
    fused.bn_act(a, x)
    a.eval()
    fused.bn_act(a, x)                                    # evaluation does not count
    fused.flush_batch_counters()
    fused.flush_batch_counters()
    assert int(a.num_batches_tracked) == 4 and not ops._PENDING_COUNTS


def test_ego_motion_planes_folded_into_the_first_temporal_block():
    """stp3.py:145-152 concatenates six broadcast ego-motion planes to the BEV features; here they enter the temporal
    model as per-frame constants (bias of the fused BatchNorms of the first block's 1x1x1 convolutions, appended to
    the whole-plane pooling).  Same outputs and gradients as the concatenated 70-channel tensor, train mode."""
    import copy
    from stp3_amd.models.temporal_model import TemporalModel
    from tests import helpers as H
    torch.manual_seed(0)
    m = H.fill_deterministic(TemporalModel(70, 3, input_shape=(20, 20), start_out_channels=64)).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m2 = copy.deepcopy(m)
    x, ego = torch.randn(2, 3, 64, 20, 20), torch.randn(2, 3, 6)
    xa = torch.cat([x, ego.view(2, 3, 6, 1, 1).expand(2, 3, 6, 20, 20)], 2).requires_grad_()
    ya = m(xa)
    ya.square().mean().backward()
    xb = x.clone().requires_grad_()
    yb = m2(xb, ego)
    yb.square().mean().backward()
    torch.testing.assert_close(yb, ya, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xb.grad, xa.grad[:, :, :64], rtol=1e-4, atol=1e-7)
    for (n, p), q in zip(m.named_parameters(), m2.parameters()):
        torch.testing.assert_close(q.grad, p.grad, rtol=1e-3, atol=1e-5, msg=n)


def test_regression_loss_with_everything_ignored_is_zero_and_differentiable():
    """losses.py:31-33: no unmasked pixel -> 0.  The masked-sum form keeps that without asking the host."""
    from stp3_amd import losses as L
    pred = torch.randn(1, 2, 2, 6, 6, requires_grad=True)
    tgt = torch.full((1, 2, 2, 6, 6), 255.0)
    loss = L.SpatialRegressionLoss(norm=1, ignore_index=255)(pred, tgt, 2)
    assert loss.item() == 0.0
    loss.backward()
    assert pred.grad.abs().max().item() == 0.0


def test_fused_loss_total_equals_the_per_term_weighting():
    """``TrainingModule._fused_total`` (what ``training_step`` returns) against the reference's per-term statement
    1 / (2 exp(w)) * L + w / 2 (trainer.py:125-172), value and gradients."""
    from stp3_amd.trainer import TrainingModule
    g = torch.Generator().manual_seed(3)
    vals = [(torch.rand((), generator=g) * 3).requires_grad_() for _ in range(7)]
    ws = [torch.nn.Parameter(torch.randn((), generator=g)) for _ in range(7)]
    per_term = sum(v / (2 * torch.exp(w)) + 0.5 * w for v, w in zip(vals, ws))
    fused = TrainingModule._fused_total(list(zip(vals, ws)))
    torch.testing.assert_close(fused, per_term, rtol=1e-6, atol=1e-6)
    ga = torch.autograd.grad(per_term, vals + ws)
    gb = torch.autograd.grad(fused, vals + ws)
    for a, b in zip(ga, gb):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)


def test_strided_data_gradient_phase_plan(monkeypatch):
    """``ops._strided_dgrad`` (one stride-1 sub-convolution per input phase instead of a convolution over the
    zero-stuffed gradient): its tap / padding / output-size planning against autograd of F.conv2d, for many kernel
    sizes, paddings, strides and ragged map sizes.  The sub-convolution launch is replaced by a float64 CPU statement
    of what stp3_conv2d_fwd computes for (top / left padding, explicit output size, zeros beyond every edge)."""
    import torch.nn.functional as F
    from stp3_amd import ops

    def launch(x, wb, bias, stride, pad, dil, out_dtype, sums_ptr=None, out_hw=None):
        assert stride == 1 and dil == (1, 1) and bias is None and out_hw is not None
        kh, kw = wb.shape[2:]
        need_h, need_w = out_hw[0] + kh - 1, out_hw[1] + kw - 1
        xp = F.pad(x.double(), (pad[1], max(need_w - pad[1] - x.shape[3], 0), pad[0], max(need_h - pad[0] - x.shape[2], 0)))
        return F.conv2d(xp[:, :, :need_h, :need_w], wb.double()).to(torch.bfloat16)

    monkeypatch.setattr(ops, '_conv2d_launch', launch)
    g = torch.Generator().manual_seed(21)
    checked = 0
    for k, pad, stride, h, w in [(7, 3, 2, 20, 21), (3, 1, 2, 9, 12), (3, 0, 2, 11, 8), (1, 0, 2, 10, 7), (5, 2, 2, 13, 13),
                                 (3, 1, 3, 10, 11), (4, 1, 2, 12, 9), (2, 0, 2, 8, 8), (5, 1, 2, 15, 10), (3, 2, 2, 7, 9)]:
        cin, cout = 3, 4
        x = torch.randn(2, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
        wgt = torch.randn(cout, cin, k, k, generator=g, dtype=torch.float64)
        y = F.conv2d(x, wgt, None, stride, pad)
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64).to(torch.bfloat16).double()
        (want,) = torch.autograd.grad(y, x, gy)
        wt = wgt.flip(2, 3).transpose(0, 1).contiguous()
        got = ops._strided_dgrad(gy, wt, x.shape, stride, (pad, pad), {})
        if got is None:                       # a phase would need negative padding: the caller zero-stuffs instead
            assert any(t is not None and t[2] < 0 for ph in range(stride) for t in [ops._phase_taps(k, pad, stride, ph)])
            continue
        checked += 1
        torch.testing.assert_close(got.double(), want, rtol=2e-2, atol=2e-2 * float(want.abs().max()))
    assert checked >= 8


def test_bench_prices_a_shape_against_its_own_roof():
    """bench.py's ``top_shapes``: a convolution whose flops per compulsory byte are below the part's ridge (2.5 PF / 8 TB/s)
    is priced in TB/s of its compulsory bytes (input + output + weights once; the weight gradient: both operands + the float32
    result), anything above it in TFLOP/s against the bf16 MFMA peak."""
    import ctypes
    import bench
    from stp3_amd import _lib, profiling
    thin = _lib.ConvDims(72, 56, 120, 32, 56, 120, 192, 1, 1, 1, 0, 0, 1, 1, 32, 192, _lib.DTYPE_BF16, 0)       # 1x1 32 -> 192
    fat = _lib.ConvDims(12, 200, 200, 128, 200, 200, 128, 3, 3, 1, 1, 1, 1, 1, 128, 128, _lib.DTYPE_BF16, 0)    # 3x3 128 -> 128
    px = 72 * 56 * 120
    assert profiling._conv_bytes([ctypes.byref(thin)]) == 2.0 * px * (32 + 192) + 2.0 * 32 * 192
    assert profiling._wgrad_bytes([ctypes.byref(thin)]) == 2.0 * px * (32 + 192) + 4.0 * 32 * 192
    for dims, want in ((thin, 'hbm'), (fat, 'mfma')):
        a = {'family': 'conv_fwd_dgrad', 'shape': 's', 'calls': 3, 'ms': 0.3, 'work': 3 * profiling._conv_flops([ctypes.byref(dims)]),
             'bytes': 3 * profiling._conv_bytes([ctypes.byref(dims)])}
        e = bench._shape_entry(a, 3)
        assert e['bound'] == want and e['calls_per_step'] == 1 and abs(e['ms_per_step'] - 0.1) < 1e-9
        roof = e['compulsory_tb_per_s'] / 8.0 if want == 'hbm' else e['tflops'] / 2500.0
        assert abs(e['frac_of_its_bound'] - roof) < 2e-3


@torch.no_grad()
def test_temporal_model_with_spatial_bottlenecks_matches_the_reference():
    """``INBETWEEN_LAYERS > 0`` (stp3/layers/temporal.py:328-375 Bottleneck3D behind every temporal block; no shipped
    configuration sets it): the reference's own TemporalModel (oracle/make_golden_inbetween.py) against the product's, same
    name-derived weights, same input -- and the same state-dict keys."""
    from oracle.make_golden_inbetween import ARGS, INPUT
    from stp3_amd.models.temporal_model import TemporalModel
    g = H.load('temporal_inbetween.npz')
    m = prep(TemporalModel(**ARGS))
    assert sorted(m.state_dict().keys()) == list(g['keys'])
    y = m(H.det_tensor(*INPUT))
    assert tuple(y.shape) == tuple(g['shape'])
    torch.testing.assert_close(H.sample(y).float(), torch.from_numpy(g['y']), rtol=2e-3, atol=2e-4)


def test_bench_line_owns_standard_output():
    import os
    """bench.py: libraries print to C stdout from their own threads (RCCL's version banner when a communicator comes up landed in
    the middle of the JSON line of a run on the N > 1 code path).  ``_claim_stdout`` points fd 1 at stderr for everybody else
    and ``_emit`` writes the line to the saved descriptor in one write: whatever else is written to fd 1 -- through Python or
    past it -- ends up on stderr, and standard output is exactly the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "fd = bench._claim_stdout()\n"
            "os.write(1, b'RCCL version : banner past Python\\n')\n"
            "print('a log line through Python')\n"
            "bench._emit(fd, {'metric': 'm', 'value': 1.5, 'config': {'workload': 'x' * 5000}})\n"
            "os.write(1, b'another banner at exit\\n')\n") % root
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, STP3_HOST_DRYRUN='1'))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0])['value'] == 1.5 and len(json.loads(lines[0])['config']['workload']) == 5000
    assert 'banner past Python' in out.stderr and 'a log line through Python' in out.stderr and 'another banner' in out.stderr
