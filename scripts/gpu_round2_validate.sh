#!/bin/bash
# First GPU visit of the next round: validate every experimental path (parity), time the kernels that changed, then
# measure which switches pay off (greedy A/B).    bash scripts/gpu_round2_validate.sh <tag> [ab]
#   part 1 (always, ~8 min): parity tests of the opt-in kernels / paths + kernel timings
#   part 2 (second argument "ab", ~12 min): scripts/ab_greedy.py -> gpurun_out/<tag>/ab_greedy.json
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { echo "=== $*"; "$@" 2>&1 | grep -v "amdgpu.ids\|MIOpen(HIP)" | tail -${TAILN:-4}; }
# -- parity ------------------------------------------------------------------------------------------------------
# opt-in kernels: conv v2 + fused conv-BN, SE passes, SE MLP, weight shadows, fused clip+Adam, BN geometry variant
TAILN=4 run env STP3_EXPERIMENTAL=1 STP3_BN_GEOM=1 timeout 300 python -m pytest tests/test_conv_v2_gpu.py tests/test_bnact_gpu.py -q -x
# voxel pool on the matrix cores (forward stage 1 and backward) against the oracle / golden vectors
TAILN=4 run env STP3_LIFT_BWD=mfma STP3_LIFT_FWD=mfma timeout 300 python -m pytest tests/test_lift_gpu.py -q -x
# BASELINE configs[4] geometry (112 KB of LDS per column)
TAILN=4 run env STP3_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_lift_stress_gpu.py -q -x
# the v2 convolution kernel as a drop-in for every bf16-output convolution (forward and data gradient)
TAILN=4 run env STP3_CONV_KERNEL=v2 timeout 300 python -m pytest tests/test_conv_gpu.py tests/test_modules_gpu.py -q -x
# the whole GPU suite through the C++ launch path
TAILN=6 run env STP3_CPP_OPS=1 timeout 400 python -m pytest tests -m gpu -q -x
# the model-level parity tests with the fused trunk operators and the new host options on
TAILN=6 run env STP3_FUSED_SE=1 STP3_SE_MLP=1 STP3_CONV_V2=1 STP3_WEIGHT_PREP=1 timeout 400 python -m pytest tests/test_modules_gpu.py -q -x
# -- kernel timings ----------------------------------------------------------------------------------------------
TAILN=10 run timeout 100 python scripts/time_lift.py
TAILN=10 run env STP3_LIFT_BWD=mfma STP3_LIFT_FWD=mfma timeout 100 python scripts/time_lift.py
TAILN=12 run env STP3_BN_GEOM=0 timeout 100 python scripts/time_bn.py
TAILN=12 run env STP3_BN_GEOM=1 timeout 100 python scripts/time_bn.py
# -- which switches pay off --------------------------------------------------------------------------------------
if [ "$2" = "ab" ]; then
  TAILN=40 run timeout 1500 python scripts/ab_greedy.py --steps 10 --warmup 3 --only bn_geom,fused_se,se_mlp,label_warp,grad_gather,weight_prep,cpp_ops,conv_kernel_v2,conv_v2,fused_adam,lazy_bn_counter,lift_mfma --out $OUT/ab_greedy.json
fi
