"""CPU: dry run of one whole training step through the GPU code path (tests/model_trace.py).

libstp3hip.so is replaced by the recording stand-in of tests/host_trace.py, tensors claim to live on the GPU and
the do-nothing kernels leave a fixed fill pattern behind, so the step runs here in seconds.  Checked: the step
executes end to end, every parameter receives a gradient of its own shape, and the C-ABI call mix is the expected one
(plan once, voxel pool once per step, one BatchNorm backward per BatchNorm layer, every dense convolution of the bf16
step on stp3_conv2d_fwd / stp3_conv2d_wgrad, ...).  Values are meaningless in a dry run; numerical parity is what the
``-m gpu`` tests establish.
"""
import collections
import os
import shutil
import subprocess
import sys

import pytest

from tests import host_trace

ROOT = host_trace.ROOT
PKG = os.path.join(ROOT, 'st-p3_amd', 'stp3_amd')
pytestmark = pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')


@pytest.fixture(scope='module')
def recorder(tmp_path_factory):
    return host_trace.build_recorder(str(tmp_path_factory.mktemp('rec') / 'libstp3hip_recorder.so'))


def _step(recorder, log):
    env = {k: v for k, v in os.environ.items() if not k.startswith('STP3_')}
    env.update(STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(log), STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'model_trace.py'), recorder], env=env, check=True,
                   timeout=600, stderr=subprocess.DEVNULL)
    with open(log) as f:
        lines = f.read().splitlines()
    assert lines[-1] == '# end'
    assert '# parameters without gradient: []' in lines and '# gradient shapes ok: True' in lines
    return lines, collections.Counter(l.split(' ', 1)[0] for l in lines if l.startswith('stp3_'))


STEPS = 2          # tests/model_trace.py runs bench.py's eager step twice (the second one after an optimizer update)


def test_step_runs_and_call_mix(recorder, tmp_path):
    _, calls = _step(recorder, tmp_path / 'default.log')
    assert calls['stp3_lift_plan_build'] == 1, calls                                # the plan is prepared once
    for per_step in ('stp3_lift_splat_fwd', 'stp3_lift_splat_bwd', 'stp3_optim_clip_adam'):
        assert calls[per_step] == STEPS, (per_step, calls[per_step])
    # one BatchNorm forward/backward pair per BatchNorm layer: the composite entry points, or -- where the convolution
    # in front produces the statistics in its epilogue -- apply-only forward and reduce + apply backward
    # (the composite ones that remain: the temporal blocks -- padded lanes, per-sample biases -- and BatchNorms on vectors)
    assert calls['stp3_bn_fwd_train'] == calls['stp3_bn_bwd_train'] > 20 * STEPS
    assert calls['stp3_bn_apply_fwd'] == calls['stp3_bn_bwd_reduce'] == calls['stp3_bn_apply_bwd'] > 70 * STEPS
    # single process: the BatchNorm passes are never split; what remains are the bias gradients of the biased convolutions
    # on the bf16 kernels (ops.channel_sums: column sums of dy through the statistics kernel), five per step
    assert calls['stp3_bn_stats'] == 5 * STEPS, calls['stp3_bn_stats']
    # the 22 MBConv blocks: depthwise -> BN1 -> swish -> squeeze-excite as ONE operator (ops_fused.dw_bn_se) -- the
    # depthwise forward with the statistics epilogue, no separate BatchNorm / pool / scale passes
    # (single process: the BatchNorm-1 constants come out of the statistics reduction itself, stp3_dwconv2d_fwd_stats_bn)
    assert calls['stp3_dwconv2d_fwd_stats_bn'] == calls['stp3_dwconv2d_bwd_data'] == calls['stp3_dwconv2d_bwd_weight_oihw'] \
        == 22 * STEPS
    assert calls['stp3_dwconv2d_fwd_stats'] == 0
    assert calls['stp3_dwconv2d_bwd_weight'] == 0                   # (the weight gradient leaves in the parameter's layout)
    assert calls['stp3_dwconv2d_fwd'] == calls['stp3_se_scale'] == 0
    # (stp3_se_pool: the five whole-plane means of a step -- pyramid pooling of the two temporal blocks, image pooling of the
    # three ASPP heads -- layers/fused.plane_mean; the squeeze of the MBConv blocks is stp3_se_pool_act)
    assert calls['stp3_se_pool'] == 5 * STEPS, calls['stp3_se_pool']
    assert calls['stp3_se_mlp_fwd'] == calls['stp3_se_mlp_bwd'] == 22 * STEPS
    for fused in ('stp3_se_pool_act', 'stp3_mbconv_scale_act', 'stp3_mbconv_bwd_reduce',
                  'stp3_mbconv_bwd_coef', 'stp3_mbconv_bwd_apply'):
        assert calls[fused] == 22 * STEPS, (fused, calls[fused])
    # expand convolution -> BN0 -> swish WITHOUT the expanded pre-activation tensor (ops_fused._PointwiseBnAct) in the
    # blocks where the streaming kernels run it (contraction <= 128 channels, >= 16384 pixels): four passes per block,
    # each its own entry point, and one stp3_bn_finalize each (the depthwise stages finish theirs inside stp3_dwconv2d_fwd_stats_bn)
    recomputed = calls['stp3_conv2d_fwd_stats']
    assert recomputed >= STEPS and recomputed % STEPS == 0, recomputed
    # (the apply pass of the recomputing blocks on the whole-row streaming kernel -- 24 -> 144, 32 -> 192: five at the full
    # image size, fewer at this one -- also writes the expand convolution's data gradient: stp3_conv2d_bn_bwd_apply_dx)
    with_dx = calls['stp3_conv2d_bn_bwd_apply_dx']
    assert STEPS <= with_dx <= 5 * STEPS and with_dx % STEPS == 0, with_dx
    assert calls['stp3_conv2d_fwd_bnact'] == calls['stp3_conv2d_bn_bwd_reduce'] == recomputed
    assert calls['stp3_conv2d_bn_bwd_apply'] + calls['stp3_conv2d_bn_bwd_apply_dx'] == recomputed
    assert calls['stp3_bn_finalize'] == recomputed, calls['stp3_bn_finalize']
    # losses and label warp on the kernels: 5 cross-entropy calls (segmentation, pedestrian, 2 HD-map elements, depth),
    # 3 regression losses, one warp launch per step
    assert calls['stp3_ce_topk_fwd'] == calls['stp3_ce_topk_bwd'] == 5 * STEPS
    assert calls['stp3_reg_loss_fwd'] == calls['stp3_reg_loss_bwd'] == 3 * STEPS
    assert calls['stp3_warp_nearest'] == STEPS
    # dense convolutions: forward + data gradient launches, one weight-gradient launch per convolution layer
    # (the split-K sum of a leaf weight's gradient is deferred: stp3_conv2d_wgrad_partials per layer, ONE
    # stp3_conv2d_wgrad_reduce_batch per step; derived weights -- merged heads, folded temporal kernels -- reduce at once)
    wgrads = calls['stp3_conv2d_wgrad'] + calls['stp3_conv2d_wgrad_partials']
    # (stp3_conv2d_fwd_add: the data gradient of the expand convolution of the MBConv blocks with an identity skip and an
    # expand layer, written together with the skip's gradient -- ops.SkipCarrier; 16 such blocks, some of them among the ones above)
    assert 16 * STEPS - with_dx <= calls['stp3_conv2d_fwd_add'] <= 16 * STEPS, calls['stp3_conv2d_fwd_add']
    assert calls['stp3_conv2d_fwd'] + calls['stp3_conv2d_fwd_add'] > 200 * STEPS and wgrads > 100 * STEPS
    # (the first step meets an arena that is too small for all of them: those layers reduce at once, the arena grows after it)
    assert calls['stp3_conv2d_wgrad_partials'] > 70 and calls['stp3_conv2d_wgrad_reduce_batch'] == STEPS, calls
    # weight shadows: once per newly met layer during the first step, then once per optimizer step -- never per use
    # (the depthwise layers' tap-major float32 copies and the weights ASSEMBLED from parameter views -- padded lanes, causal taps
    # side by side, merged heads, split projections: ops.assembled_weight -- are rows of the same table)
    n_layers = calls['stp3_conv2d_prep_weights'] - STEPS
    n_depthwise = calls['stp3_dwconv2d_fwd_stats_bn'] // STEPS
    assert 0 < n_layers <= wgrads // STEPS + 1 + n_depthwise, (n_layers, wgrads, n_depthwise)
    # ... and the gradients of the assembled weights go back to their parameters in ONE launch per backward pass
    assert calls['stp3_conv2d_scatter_weight_grads'] == STEPS, calls['stp3_conv2d_scatter_weight_grads']
