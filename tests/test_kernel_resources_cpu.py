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
        assert k['scratch'] == 0 and k['sgpr_spills'] == 0, k               # no scratch memory, no spills of any kind
    # the memory-bound kernels keep at least 3 waves per SIMD resident (<= 168 registers); the voxel pool's fit 4 and 3
    for name, budget in (('lift_column_mma_kernel', 128), ('void lift_column_kernel<12>', 96), ('void lift_gather_kernel<float, 3>', 64), ('void lift_gather_kernel<unsigned short, 3>', 64), ('void lift_gather_kernel<unsigned short, 8>', 64),
                         ('lift_bwd_kernel', 168), ('plan_columns_kernel', 64), ('plan_fill_kernel', 64),
                         ('prep_weights_kernel', 128), ('optim_sumsq_kernel', 128), ('optim_prepare_kernel', 128),
                         ('optim_adam_kernel', 128), ('void se_mlp_fwd_kernel<40>', 128), ('void se_mlp_bwd_sample_kernel<40>', 128),
                         ('void se_mlp_fwd_kernel<8>', 128), ('void se_mlp_bwd_sample_kernel<16>', 128),
                         ('se_mlp_bwd_weight_kernel', 128), ('voxels_sum_fwd_kernel', 128),
                         ('voxels_sum_bwd_kernel', 128), ('void dwconv_fwd_stats_kernel<unsigned short, 3, 1, unsigned int>', 168),
                         ('void dwconv_fwd_stats_kernel<unsigned short, 5, 1, unsigned int>', 256),
                         ('void mbconv_bwd_reduce_kernel<unsigned short, 8, 2>', 128),
                         ('void mbconv_bwd_apply_kernel<unsigned short, 8, 2>', 168),
                         # (1024 threads per workgroup: at most 128 registers; the row's 40 losses per thread live in them)
                         ('void topk_select_kernel<true>', 128), ('void topk_select_kernel<false>', 64),
                         # the batched gate kernels: every phase's loads in flight at once, one workgroup of 4 waves per sample
                         ('void se_mlp_fwd_batched_kernel<40>', 256), ('void se_mlp_bwd_sample_batched_kernel<40>', 256)):
        assert name in by_name, name
        for k in by_name[name]:
            assert k['vgpr'] + k['agpr'] <= budget, k
    # the MFMA convolutions: accumulators in AGPRs, at least 2 workgroups of 256 threads per CU
    for prefix in ('void conv2d_igemm_kernel<128, 0, 0, ', 'void conv2d_wgrad_kernel<128, 128, '):
        found = [k for n, ks in by_name.items() if n.startswith(prefix) for k in ks]
        assert found, prefix
        for k in found:
            assert k['agpr'] == 64 and k['vgpr'] + k['agpr'] <= 256, k
    # The grids of the streaming kernels are sized to ONE resident round of the chip (256 CUs x the workgroups a CU keeps):
    # the launchers carry these numbers as constants (plan() in stp3_bnact.hip, mb_plan() in stp3_mbconv.hip, wgrad_plan() in
    # stp3_conv.hip) -- a kernel that grows beyond its register budget would silently get a second, nearly empty round
    def per_cu(k):
        regs = (k['vgpr'] + k['agpr'] + 7) // 8 * 8
        return min(8, 512 // regs)
    for prefix, want in (('void bn_stats_kernel<unsigned short, 8', 7), ('void bn_apply_fwd_kernel<unsigned short, 8', 5),
                         ('void bn_bwd_reduce_kernel<unsigned short, 8', 4), ('void bn_apply_bwd_kernel<unsigned short, 8', 3),
                         ('void se_pool_act_kernel<unsigned short, 8', 7), ('void mbconv_scale_act_kernel<unsigned short, 8', 7),
                         ('void mbconv_bwd_reduce_kernel<unsigned short, 8', 4), ('void mbconv_bwd_apply_kernel<unsigned short, 8', 4),
                         ('void conv2d_wgrad_kernel<128, 128, ', 2), ('void conv2d_wgrad_kernel<128, 64, ', 3),
                         ('void conv2d_wgrad_kernel<64, 128, ', 3), ('void conv2d_wgrad_kernel<64, 64, ', 5)):
        found = [k for n, ks in by_name.items() if n.startswith(prefix) for k in ks]
        assert found, prefix
        for k in found:
            assert per_cu(k) >= want, (want, k)
    # the streaming pointwise kernels: whole-row variant 2-3 workgroups per CU (launch bound), register-only variant 4 for the
    # thin layers
    for n, ks in by_name.items():
        if n.startswith('void pointwise_rows_kernel<'):
            assert all(k['vgpr'] + k['agpr'] <= 256 for k in ks), n
        if n.startswith('void pointwise_direct_kernel<2, 0>') or n.startswith('void pointwise_direct_kernel<2, 1>'):
            assert all(per_cu(k) >= 4 for k in ks), n
