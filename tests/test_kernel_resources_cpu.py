"""CPU: static resources of the gfx950 build (scripts/kernel_resources.py: hipcc cross-compiles without a GPU).

What an execution of the kernels on the CPU cannot see: register spills to scratch, static LDS against the launch
limit, register budgets.  The values are facts of the build; the assertions fix the properties the kernels rely on."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc') or shutil.which('c++filt') is None, reason='needs hipcc')
def test_kernel_register_lds_and_scratch_budgets():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'kernel_resources.py'), '--json'],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-1500:]
    rows = json.loads(out.stdout)
    assert len(rows) > 80                                               # every template instantiation is a kernel
    by_name = {}
    for k in rows:
        by_name.setdefault(k['kernel'], []).append(k)
        assert k['vgpr'] + k['agpr'] <= 512, k                           # the unified register file of a wave
        assert k['lds_static'] <= 64 * 1024, k                           # static LDS: default launch limit
        assert k['vgpr_spills'] == 0, k                                  # no vector register spills anywhere
        assert k['scratch'] <= 64, k                                     # (the MFMA convolution keeps a 32-byte array there)
    # the kernels that have not run on hardware yet: no scratch, no spills of any kind, at least 4 waves per SIMD
    for name in ('lift_runs_mfma_kernel', 'lift_bwd_mfma_kernel', 'prep_weights_kernel', 'optim_sumsq_kernel',
                 'optim_prepare_kernel', 'optim_adam_kernel', 'se_mlp_fwd_kernel', 'se_mlp_bwd_sample_kernel',
                 'se_mlp_bwd_weight_kernel', 'voxels_sum_fwd_kernel', 'voxels_sum_bwd_kernel'):
        assert name in by_name, name
        for k in by_name[name]:
            assert k['scratch'] == 0 and k['sgpr_spills'] == 0 and k['vgpr'] + k['agpr'] <= 128, k
