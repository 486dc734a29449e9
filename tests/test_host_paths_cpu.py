"""CPU: the two launch paths of the custom operators hand the C ABI exactly the same calls.

The Python/ctypes path (stp3_amd/ops.py) is the one the GPU parity tests validate; the C++ path
(csrc/host/stp3_host.cpp, STP3_CPP_OPS=1) must drive the same kernels the same way.  tests/host_trace.py builds a
recording stand-in for libstp3hip.so from include/stp3_hip.h and pushes a fixed set of operator calls (fused
BatchNorm variants, dense convolutions with their data / weight gradients, depthwise convolutions; forward and
backward) through ``stp3_amd.ops``; the recorded traces -- entry point, dims struct, scalars, null / aliasing
pattern of the pointers, checksums of the caller's input buffers, shapes / dtypes / strides of what autograd
returns -- have to be identical line by line.
"""
import os
import shutil
import subprocess
import sys

import pytest

from tests import host_trace

ROOT = host_trace.ROOT
PKG = os.path.join(ROOT, 'st-p3_amd', 'stp3_amd')


def _trace(recorder, log, cpp):
    env = dict(os.environ, STP3_CPP_OPS='1' if cpp else '0', STP3_HOST_DRYRUN='1', STP3_TRACE_LOG=log,
               STP3_REAL_LIB=os.path.join(PKG, 'libstp3hip.so'))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'host_trace.py'), recorder], env=env, check=True,
                   timeout=300)
    with open(log) as f:
        return f.read().splitlines()


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc for the recording library')
@pytest.mark.skipif(not os.path.exists(os.path.join(PKG, '_stp3_host.so')), reason='C++ launch path not built')
def test_cpp_and_python_launch_paths_issue_identical_calls(tmp_path):
    recorder = host_trace.build_recorder(str(tmp_path / 'libstp3hip_recorder.so'))
    py = _trace(recorder, str(tmp_path / 'python.log'), cpp=False)
    cpp = _trace(recorder, str(tmp_path / 'cpp.log'), cpp=True)
    calls = [l for l in py if l.startswith('stp3_')]
    assert len(calls) > 90, 'the driver did not reach the library'
    for name in ('stp3_bn_fwd_train', 'stp3_bn_bwd_train', 'stp3_bn_apply_fwd', 'stp3_bn_bwd_reduce',
                 'stp3_bn_apply_bwd', 'stp3_conv2d_fwd', 'stp3_conv2d_wgrad', 'stp3_dwconv2d_fwd',
                 'stp3_dwconv2d_bwd_data', 'stp3_dwconv2d_bwd_weight'):
        assert any(l.startswith(name + ' ') for l in calls), name
    assert py[-1] == '# end' and cpp[-1] == '# end'
    diff = [(i, a, b) for i, (a, b) in enumerate(zip(py, cpp)) if a != b]
    assert not diff, f'first difference at line {diff[0][0]}:\n  python: {diff[0][1]}\n  c++   : {diff[0][2]}'
    assert len(py) == len(cpp)
