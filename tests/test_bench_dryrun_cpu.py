"""CPU: bench.py's main() end to end as a dry run -- the script the driver runs on the MI355X must not fall over on
a Python-level error.  The GPU code path runs on CPU tensors against the recording stand-in for libstp3hip.so
(tests/model_trace.py); B=1 of the real 6-camera 224x480 x T=3 workload, one warm-up and one timed step, the
voxel-pool roofline leg included (event timing stubbed); the cpu_baseline leg is a subprocess of the same file and
has its own flag.  Checks the one-line JSON contract (keys, types, the workload named after BASELINE.json)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from tests import host_trace

ROOT = host_trace.ROOT
PKG = os.path.join(ROOT, 'st-p3_amd', 'stp3_amd')


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')
@pytest.mark.parametrize('workload', ['c3', 'perception', 'prediction', 'planning'])
def test_bench_main_dry_run(tmp_path, workload):
    recorder = host_trace.build_recorder(str(tmp_path / 'libstp3hip_recorder.so'))
    env = {k: v for k, v in os.environ.items() if not k.startswith('STP3_')}
    env.update(STP3_BENCH_DRYRUN='1', STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(tmp_path / 'trace.log'),
               STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_dryrun.py'), recorder, '--steps', '1',
                          '--warmup', '1', '--batch', '1', '--no-cpu-baseline', '--workload', workload],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'falling back' not in out.stderr, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                        # ONE JSON line
    line = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 1 and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['unit'] == 'samples/s' and line['higher_is_better'] is True and line['data'] == 'synthetic'
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert ('depth CE' in line['config']['workload']) == (workload == 'c3')
    assert ('future frames' in line['metric']) == (workload in ('prediction', 'planning'))
    assert line['config']['host_options'] == 'grad_gather=1 label_warp=batched lazy_bn_counter=1'   # bit-identical host options
    roof = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in roof, key
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    # B=1: 3 frames x (feat + depth probabilities + BEV planes) float32, SURVEY.md section 8d: 14 755 840 B per frame
    assert roof['algorithmic_bytes_per_launch'] == 3 * 14755840
    # what ships is what runs: every operator family of the C ABI shows up in the call trace of one step (the plain
    # stp3_dwconv2d_fwd / stp3_se_pool / stp3_se_scale serve the evaluation-mode forward: BatchNorm on running statistics)
    trace = open(tmp_path / 'trace.log').read()
    if workload == 'planning':
        assert 'stp3_traj_cost_fwd ' in trace and 'stp3_traj_cost_bwd ' in trace
    if workload in ('prediction', 'planning'):                # the prediction stage's own operators
        for entry in ('stp3_dwconv2d_fwd_bias', 'stp3_layernorm_fwd', 'stp3_layernorm_bwd', 'stp3_gru_reset_cat_fwd',
                      'stp3_gru_output_fwd', 'stp3_gru_output_bwd', 'stp3_gru_reset_cat_bwd'):
            assert entry + ' ' in trace, entry
    for entry in ('stp3_lift_plan_build', 'stp3_lift_splat_fwd', 'stp3_lift_splat_bwd',
                  'stp3_conv2d_fwd', 'stp3_conv2d_fwd_add', 'stp3_conv2d_bn_bwd_apply_dx', 'stp3_conv2d_wgrad', 'stp3_conv2d_wgrad_partials',
                  'stp3_conv2d_wgrad_reduce_batch',
                  'stp3_conv2d_prep_weights', 'stp3_bn_fwd_train',
                  'stp3_dwconv2d_fwd_stats_bn', 'stp3_bn_finalize', 'stp3_se_pool_act', 'stp3_se_mlp_fwd',
                  'stp3_se_mlp_bwd', 'stp3_mbconv_scale_act', 'stp3_mbconv_bwd_reduce', 'stp3_mbconv_bwd_coef',
                  'stp3_mbconv_bwd_apply',
                  'stp3_dwconv2d_bwd_data', 'stp3_dwconv2d_bwd_weight_oihw', 'stp3_optim_clip_adam',
                  'stp3_ce_topk_fwd', 'stp3_ce_topk_bwd', 'stp3_warp_nearest', 'stp3_linear_fwd', 'stp3_linear_bwd',
                  'stp3_sum_n_plane', 'stp3_se_pool',
                  'stp3_causal_pair_fwd', 'stp3_causal_pair_bwd', 'stp3_upsample_bilinear_fwd', 'stp3_upsample_bilinear_bwd'):
        assert entry + ' ' in trace, entry
    if workload == 'c3':                                      # the instance / flow regression losses exist in c3 only
        assert 'stp3_reg_loss_fwd ' in trace and 'stp3_reg_loss_bwd ' in trace


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher around it: bench.py starts torch.distributed.run itself (one rank
    per GPU, rendezvous on 127.0.0.1) and the job prints the one whole-job JSON line."""
    recorder = host_trace.build_recorder(str(tmp_path / 'libstp3hip_recorder.so'))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(STP3_BENCH_DRYRUN='1', STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(tmp_path / 'trace.log'),
               STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_dryrun.py'), recorder, '--gpus', '2',
                          '--steps', '1', '--warmup', '1', '--batch', '1', '--no-cpu-baseline', '--no-roofline'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['parallelism'] == 'dp2' and line['config']['global_batch'] == 2


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')
def test_bench_main_dry_run_two_ranks(tmp_path):
    """The driver's N>1 launch line (torch.distributed.run, one rank per GPU) with the gloo backend: barrier,
    max-over-ranks timing, cross-replica BatchNorm statistics and bucketed gradient all-reduce all execute; rank 0
    prints the one JSON line with the whole-job value."""
    import socket
    recorder = host_trace.build_recorder(str(tmp_path / 'libstp3hip_recorder.so'))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, STP3_BENCH_DRYRUN='1', STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(tmp_path / 'trace.log'),
               STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(ROOT, 'tests', 'bench_dryrun.py'), recorder, '--gpus', '2', '--steps', '1',
                          '--warmup', '1', '--batch', '1', '--no-cpu-baseline', '--no-roofline'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                        # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 2 and line['config']['parallelism'] == 'dp2'
    assert abs(line['value'] - 2 * 1000.0 / line['ms_per_step']) < 1e-2 * line['value'] + 1e-3   # whole-job rate
    calls = [l for l in open(tmp_path / 'trace.log') if l.startswith('stp3_bn_')]
    # more than one rank: the BatchNorm operator runs split (statistics | all-reduce | apply), never the composite
    assert any(l.startswith('stp3_bn_stats ') for l in calls) and not any(l.startswith('stp3_bn_fwd_train ') for l in calls)


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')
def test_bench_fails_when_its_first_step_breaks(tmp_path):
    """No fallback ladder: a failure in the first step ends the bench with a non-zero exit code and NO JSON line -- never
    a line that quietly carries another workload or other host options."""
    recorder = host_trace.build_recorder(str(tmp_path / 'libstp3hip_recorder.so'))
    env = {k: v for k, v in os.environ.items() if not k.startswith('STP3_')}
    env.update(STP3_BENCH_DRYRUN='1', STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(tmp_path / 'trace.log'),
               STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'), STP3_TEST_BREAK_GATHER='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_dryrun.py'), recorder, '--steps', '1',
                          '--warmup', '1', '--batch', '1', '--no-cpu-baseline', '--no-roofline'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert 'falling back' not in out.stderr


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')
def test_bench_eight_ranks_dry_run(tmp_path):
    """The line the driver runs on an 8-GPU node -- ``torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`` -- with
    EIGHT processes here (gloo, CPU tensors, the recording stand-in for the kernels): rank / environment handling, the
    barrier + max-over-ranks timing, 8-way bucketed gradient all-reduce, cross-replica BatchNorm statistics with the
    sibling layers sharing their exchanges, the per-rank clocks and the collective counts in the one JSON line.  Nobody
    has run this job on eight GPUs yet; this is everything about it that can be executed without them."""
    import socket
    recorder = host_trace.build_recorder(str(tmp_path / 'libstp3hip_recorder.so'))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, STP3_BENCH_DRYRUN='1', STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=str(tmp_path / 'trace.log'),
               STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'), OMP_NUM_THREADS='1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8',
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(ROOT, 'tests', 'bench_dryrun.py'), recorder, '--gpus', '8', '--steps', '2',
                          '--warmup', '1', '--batch', '1', '--no-cpu-baseline', '--no-roofline'],
                         env=env, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                        # rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['config']['global_batch'] == 8 and line['config']['parallelism'] == 'dp8'
    assert line['scaling'] == 'weak' and line['steps'] == 2
    assert abs(line['value'] - 8 * 1000.0 / line['ms_per_step']) < 1e-2 * line['value'] + 1e-3     # whole-job rate
    assert len(line['per_rank_ms_per_step']) == 8 and abs(max(line['per_rank_ms_per_step']) - line['ms_per_step']) < 1e-2
    coll = line['collectives_per_step']
    # 129 train-mode BatchNorm layers with the six heads of configs[2]; the sibling layers share their exchange:
    # 103 forward + 103 backward all-reduces per step, and a handful of gradient buckets
    assert coll['batchnorm_statistics_all_reduces'] == 206, coll
    assert 1 <= coll['gradient_bucket_all_reduces'] <= 16, coll
