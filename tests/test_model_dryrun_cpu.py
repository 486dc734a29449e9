"""CPU: dry run of one whole training step through the GPU code path (tests/model_trace.py).

libstp3hip.so is replaced by the recording stand-in of tests/host_trace.py, tensors claim to live on the GPU and
the do-nothing kernels leave a fixed fill pattern behind, so the step runs here in seconds.  Checked:
  * the default path executes end to end, every parameter receives a gradient of its own shape, and the C-ABI
    call mix is the expected one (lift path once, one BatchNorm forward/backward pair per BatchNorm layer, ...);
  * the C++ launch path (STP3_CPP_OPS=1) makes exactly the same calls with the same arguments for the whole step;
  * every experimental switch combination that scripts/gpu_round2_validate.sh A/Bs on the MI355X executes end to
    end (so a GPU visit is not spent on a Python-level error) and routes work to the entry points it claims to.
Values are meaningless in a dry run; numerical parity is what the ``-m gpu`` tests establish.
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
FLAGS = ('STP3_CPP_OPS', 'STP3_BN_GEOM', 'STP3_FUSED_SE', 'STP3_CONV_V2', 'STP3_MFMA_CONV', 'STP3_LIFT_BWD',
         'STP3_WEIGHT_PREP', 'STP3_GRAD_GATHER', 'STP3_LABEL_WARP', 'STP3_FUSED_ADAM', 'STP3_LAZY_BN_COUNTER',
         'STP3_LIFT_FWD', 'STP3_SE_MLP')

pytestmark = pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')


@pytest.fixture(scope='module')
def recorder(tmp_path_factory):
    return host_trace.build_recorder(str(tmp_path_factory.mktemp('rec') / 'libstp3hip_recorder.so'))


def _step(recorder, log, **flags):
    env = {k: v for k, v in os.environ.items() if k not in FLAGS}
    env.update(flags, STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(log), STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'model_trace.py'), recorder], env=env, check=True,
                   timeout=600, stderr=subprocess.DEVNULL)
    with open(log) as f:
        lines = f.read().splitlines()
    assert lines[-1] == '# end'
    assert '# parameters without gradient: []' in lines and '# gradient shapes ok: True' in lines
    return lines, collections.Counter(l.split(' ', 1)[0] for l in lines if l.startswith('stp3_'))


STEPS = 2          # tests/model_trace.py runs bench.py's eager step twice (the second one after an optimizer update)


def test_default_step_runs_and_call_mix(recorder, tmp_path):
    _, calls = _step(recorder, tmp_path / 'default.log')
    for once in ('stp3_voxel_index', 'stp3_lift_plan_build'):                    # the plan is prepared once
        assert calls[once] == 1, (once, calls[once])
    for per_step in ('stp3_depth_softmax', 'stp3_lift_splat_fwd', 'stp3_lift_splat_bwd'):
        assert calls[per_step] == STEPS, (per_step, calls[per_step])
    assert calls['stp3_bn_fwd_train'] == calls['stp3_bn_bwd_train'] >= 100 * STEPS   # one pair per BatchNorm layer
    assert calls['stp3_dwconv2d_fwd'] == calls['stp3_dwconv2d_bwd_data'] == calls['stp3_dwconv2d_bwd_weight'] \
        == 22 * STEPS
    assert calls['stp3_conv2d_fwd'] > 100 * STEPS and calls['stp3_conv2d_wgrad'] > 0
    assert not any(k.startswith(('stp3_se_', 'stp3_conv2d_fwd_v2', 'stp3_conv2d_prep')) for k in calls)   # switches off


@pytest.mark.skipif(not os.path.exists(os.path.join(PKG, '_stp3_host.so')), reason='C++ launch path not built')
def test_cpp_launch_path_makes_the_same_calls_for_the_whole_step(recorder, tmp_path):
    py, _ = _step(recorder, tmp_path / 'python.log')
    cpp, _ = _step(recorder, tmp_path / 'cpp.log', STP3_CPP_OPS='1')
    diff = [(i, a, b) for i, (a, b) in enumerate(zip(py, cpp)) if a != b]
    assert not diff, f'first difference at line {diff[0][0]}:\n  python: {diff[0][1]}\n  c++   : {diff[0][2]}'
    assert len(py) == len(cpp) > 1000


@pytest.mark.parametrize('name,flags,expect', [
    ('bngeom', dict(STP3_BN_GEOM='1'), ()),
    ('se', dict(STP3_FUSED_SE='1'), ('stp3_se_pool', 'stp3_se_scale')),
    ('convv2', dict(STP3_CONV_V2='1'), ('stp3_conv2d_fwd_v2',)),
    ('all', dict(STP3_BN_GEOM='1', STP3_FUSED_SE='1', STP3_CONV_V2='1'), ('stp3_se_pool', 'stp3_conv2d_fwd_v2')),
    ('trunkfused', dict(STP3_BN_GEOM='1', STP3_FUSED_SE='1', STP3_CONV_V2='1', STP3_MFMA_CONV='all'),
     ('stp3_se_pool', 'stp3_conv2d_fwd_v2')),
    ('mfma_all', dict(STP3_MFMA_CONV='all'), ()),
    ('mfma_off', dict(STP3_MFMA_CONV='0'), ()),
    ('weight_prep', dict(STP3_WEIGHT_PREP='1'), ('stp3_conv2d_prep_weights',)),
    ('everything', dict(STP3_BN_GEOM='1', STP3_FUSED_SE='1', STP3_CONV_V2='1', STP3_MFMA_CONV='all',
                        STP3_WEIGHT_PREP='1', STP3_GRAD_GATHER='1', STP3_LABEL_WARP='batched', STP3_FUSED_ADAM='1',
                        STP3_LAZY_BN_COUNTER='1', STP3_LIFT_FWD='mfma', STP3_LIFT_BWD='mfma', STP3_SE_MLP='1'),
     ('stp3_conv2d_prep_weights', 'stp3_se_pool', 'stp3_optim_clip_adam', 'stp3_se_mlp_fwd', 'stp3_se_mlp_bwd')),
])
def test_experimental_switches_run_end_to_end(recorder, tmp_path, name, flags, expect):
    _, calls = _step(recorder, tmp_path / f'{name}.log', **flags)
    for entry in expect:
        assert calls[entry] > 0, (name, entry)
    if 'STP3_WEIGHT_PREP' in flags:
        # once per newly met layer during the first step, then once per optimizer step -- never once per use
        n_layers = calls['stp3_conv2d_prep_weights'] - STEPS
        assert 0 < n_layers < calls['stp3_conv2d_fwd'] + calls['stp3_conv2d_fwd_v2']
    if name == 'mfma_off':
        assert calls['stp3_conv2d_fwd'] == 0
    if name == 'trunkfused':
        assert calls['stp3_conv2d_fwd_v2'] > 80 * STEPS          # the EfficientNet trunk's expand / project convolutions too
