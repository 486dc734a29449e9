"""GPU: the dense modules and the whole STP3.forward against fixtures generated from the
reference's own classes on the CPU (oracle/make_golden_modules.py -> tests/golden/modules.npz).

Tolerances: float32 on the GPU vs float32 on the CPU, different conv algorithms and summation
orders: rtol 2e-3 / atol 2e-4 on activations of O(0.1-1).  bf16 autocast: rtol 5e-2 / atol 3e-2
(2e-2 on activations) and segmentation IoU within 1e-3 (BASELINE.json).
"""
import numpy as np
import pytest
import torch

from stp3_amd import synthetic
from stp3_amd.config import perception_cfg
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = None


def golden():
    global G
    if G is None:
        G = H.load('modules.npz')
    return G


def close(actual, key, rtol=2e-3, atol=2e-4):
    torch.testing.assert_close(H.sample(actual).float().cpu(), torch.from_numpy(golden()[key]), rtol=rtol, atol=atol)


def prep(m):
    return H.fill_deterministic(m).eval().cuda()


@torch.no_grad()
def test_heads_and_upsampling():
    from stp3_amd.layers.convolutions import DeepLabHead, UpsamplingAdd, UpsamplingConcat
    close(prep(DeepLabHead(160, 160, hidden_channel=64))(H.det_tensor((2, 160, 14, 30), 1).cuda()), 'deeplab_enc')
    close(prep(DeepLabHead(64, 64, hidden_channel=128))(H.det_tensor((1, 64, 200, 200), 2).cuda()), 'deeplab_bev')
    close(prep(UpsamplingConcat(216, 64))(H.det_tensor((2, 160, 14, 30), 3).cuda(),
                                          H.det_tensor((2, 56, 28, 60), 4).cuda()), 'upconcat')
    close(prep(UpsamplingAdd(256, 128))(H.det_tensor((2, 256, 25, 25), 5).cuda(),
                                        H.det_tensor((2, 128, 50, 50), 6).cuda()), 'upadd')


@torch.no_grad()
def test_temporal_block_and_model():
    from stp3_amd.layers.temporal import TemporalBlock
    from stp3_amd.models.temporal_model import TemporalModel
    m = prep(TemporalBlock(70, 64, use_pyramid_pooling=True, pool_sizes=[(2, 40, 40)]))
    close(m(H.det_tensor((2, 70, 3, 40, 40), 7).cuda()), 'tblock')
    m = prep(TemporalModel(70, 3, input_shape=(200, 200), start_out_channels=64))
    close(m(H.det_tensor((1, 3, 70, 200, 200), 8).cuda()), 'tmodel')


@torch.no_grad()
def test_decoder_and_encoder():
    from stp3_amd.models.decoder import Decoder
    from stp3_amd.models.encoder import Encoder
    gate = {'perceive_hdmap': True, 'predict_pedestrian': True, 'predict_instance': False,
            'predict_future_flow': False, 'planning': False}
    o = prep(Decoder(64, 2, 3, 2, gate))(H.det_tensor((1, 3, 64, 200, 200), 9).cuda())
    for k in ('segmentation', 'pedestrian', 'hdmap'):
        close(o[k], f'decoder_{k}')
    assert o['instance_center'] is None and o['costvolume'] is None
    f, d = prep(Encoder(perception_cfg().MODEL.ENCODER, D=48))(H.det_tensor((2, 3, 224, 480), 10).cuda())
    close(f, 'encoder_feat')
    close(d, 'encoder_depth')


def _iou_counts(seg_logits, batch):
    pr = seg_logits.argmax(dim=2)[:, 2:].cpu()
    tgt = batch['segmentation'][:, 2:, 0]
    return np.array([int(((pr == 1) & (tgt == 1)).sum()), int(((pr == 1) & (tgt == 0)).sum()),
                     int(((pr == 0) & (tgt == 1)).sum())])


def _iou(c):
    return c[0] / max(1, c.sum())


@torch.no_grad()
def test_full_forward_fp32_and_bf16_iou():
    """BASELINE.json configs[1]-style check at B=1: full STP3.forward (encoder + HIP lift + temporal +
    decoder) vs the reference class run on the CPU with the same weights and inputs."""
    from stp3_amd.models.stp3 import STP3
    from stp3_amd.metrics import IntersectionOverUnion
    model = prep(STP3(perception_cfg()))
    batch = synthetic.make_batch(batch=1, seq=3, seed=2)
    img = batch['image'].cuda()
    o = model(img, batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'])
    for k in ('segmentation', 'pedestrian', 'hdmap', 'depth_prediction'):
        close(o[k], f'stp3_{k}', rtol=5e-3, atol=1e-3)
    ref_counts = golden()['stp3_iou_counts']
    assert abs(_iou(_iou_counts(o['segmentation'], batch)) - _iou(ref_counts)) <= 1e-3
    # the product metric class computes the same IoU from the same counts
    metric = IntersectionOverUnion(2).cuda()
    metric(o['segmentation'].argmax(dim=2, keepdim=True)[:, 2:], batch['segmentation'][:, 2:].cuda())
    assert abs(metric.compute()[1].item() - _iou(_iou_counts(o['segmentation'], batch))) < 1e-6
    # pose tensors may also arrive on the GPU (reference behaviour): same result
    o2 = model(img, batch['intrinsics'].cuda(), batch['extrinsics'].cuda(), batch['future_egomotion'].cuda())
    # (float32 runs on the library's own split-MFMA convolutions, fixed summation orders everywhere: the same bits)
    assert torch.equal(o2['segmentation'], o['segmentation'])
    # bf16 convolutions (the benchmarked precision), channels-last
    from stp3_amd.utils import to_channels_last
    model = to_channels_last(model)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ob = model(img, batch['intrinsics'], batch['extrinsics'], batch['future_egomotion'])
    close(ob['segmentation'], 'stp3_segmentation', rtol=5e-2, atol=3e-2)
    assert abs(_iou(_iou_counts(ob['segmentation'].float(), batch)) - _iou(ref_counts)) <= 1e-3


def test_losses_and_label_warp_on_gpu():
    from stp3_amd import geometry as geo
    from stp3_amd import losses as L
    g = golden()
    pred = H.det_tensor((2, 3, 2, 200, 200), 11, 3.0).cuda()
    seg, ped, hd = synthetic.make_labels(2, 3, seed=4)
    l1 = L.SegmentationLoss(torch.Tensor([1.0, 2.0]), use_top_k=True, top_k_ratio=0.25, future_discount=0.95)(
        pred, seg.cuda(), 3)
    l2 = L.HDmapLoss(torch.Tensor([[1.0, 5.0], [1.0, 1.0]]), [1, 1], [True, False], [0.25, 0.25])(
        H.det_tensor((2, 4, 200, 200), 12, 3.0).cuda(), hd[:, 2].cuda())
    l3 = L.DepthLoss()(H.det_tensor((1, 2, 2, 48, 28, 60), 13, 3.0).cuda(),
                       (H.det_tensor((1, 2, 2, 28, 60), 14).abs() * 47).long().cuda())
    tgt = H.det_tensor((2, 3, 2, 50, 50), 15)
    tgt[:, :, :, :10] = 255
    l4 = L.SpatialRegressionLoss(norm=1, future_discount=0.95)(H.det_tensor((2, 3, 2, 50, 50), 16).cuda(),
                                                               tgt.cuda(), 2)
    got = torch.stack([l1, l2, l3, l4]).double().cpu()
    torch.testing.assert_close(got, torch.from_numpy(g['losses']), rtol=1e-5, atol=1e-6)
    ego = synthetic.make_rig(2, 3, seed=6)[2]
    wp = geo.cumulative_warp_features(seg.float().cuda(), ego.cuda(), 'nearest', (50.0, 50.0))
    wr = geo.cumulative_warp_features_reverse(seg.float().cuda(), ego.cuda(), 'nearest', (50.0, 50.0))
    # nearest sampling on the GPU may round a handful of border samples differently from the CPU
    assert (wp.sum(dim=(-1, -2, -3)).cpu() - torch.from_numpy(g['warp_past_sum'])).abs().max() <= 64
    assert (wr.sum(dim=(-1, -2, -3)).cpu() - torch.from_numpy(g['warp_rev_sum'])).abs().max() <= 64
    assert (H.sample(wp).cpu() != torch.from_numpy(g['warp_past_sample'])).float().mean() < 2e-3
