#!/bin/bash
# First GPU visit of the next round: validate every experimental path, then A/B them in the bench (short runs).
#   bash scripts/gpu_round2_validate.sh <tag>
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { echo "=== $*"; "$@" 2>&1 | grep -v "amdgpu.ids\|MIOpen(HIP)" | tail -${TAILN:-4}; }
TAILN=4 run env STP3_EXPERIMENTAL=1 STP3_BN_GEOM=1 timeout 200 python -m pytest tests/test_conv_v2_gpu.py tests/test_bnact_gpu.py -q -x
TAILN=4 run env STP3_LIFT_BWD=mfma STP3_LIFT_FWD=mfma timeout 300 python -m pytest tests/test_lift_gpu.py -q -x
TAILN=4 run env STP3_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_lift_stress_gpu.py -q -x
TAILN=10 run env STP3_LIFT_BWD=mfma STP3_LIFT_FWD=mfma timeout 100 python scripts/time_lift.py
TAILN=6 run env STP3_CPP_OPS=1 timeout 400 python -m pytest tests -m gpu -q -x
TAILN=12 run env STP3_BN_GEOM=0 timeout 100 python scripts/time_bn.py
TAILN=12 run env STP3_BN_GEOM=1 timeout 100 python scripts/time_bn.py
bench() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload ${WL:-c3} > $OUT/bench_$name.log 2> $OUT/bench_$name.err
          echo "bench $name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.log) $(grep -v "amdgpu.ids\|MIOpen(HIP)" $OUT/bench_$name.err | tail -2 | cut -c1-200)"; }
bench base STP3_BN_GEOM=0
bench plainhost STP3_GRAD_GATHER=0 STP3_LABEL_WARP=per_label STP3_LAZY_BN_COUNTER=0     # bench.py opts into these by default
WL=perception bench base_perception STP3_BN_GEOM=0     # continuity with the round-1 numbers (88.5 ms/step)
bench bngeom STP3_BN_GEOM=1
bench se STP3_BN_GEOM=1 STP3_FUSED_SE=1
bench semlp STP3_BN_GEOM=1 STP3_FUSED_SE=1 STP3_SE_MLP=1
bench convv2 STP3_BN_GEOM=1 STP3_FUSED_SE=1 STP3_CONV_V2=1
bench liftmfma STP3_BN_GEOM=0 STP3_LIFT_BWD=mfma STP3_LIFT_FWD=mfma
bench mfmaall STP3_MFMA_CONV=all
bench wgrad64 STP3_WGRAD_MIN_CHANNELS=64
bench wprep STP3_WEIGHT_PREP=1
bench fusedadam STP3_FUSED_ADAM=1
bench cpp STP3_BN_GEOM=1 STP3_CPP_OPS=1
bench trunkfused STP3_BN_GEOM=1 STP3_FUSED_SE=1 STP3_CONV_V2=1 STP3_MFMA_CONV=all
bench all STP3_BN_GEOM=1 STP3_FUSED_SE=1 STP3_CONV_V2=1 STP3_MFMA_CONV=all STP3_WEIGHT_PREP=1 STP3_GRAD_GATHER=1 STP3_LIFT_BWD=mfma STP3_LIFT_FWD=mfma STP3_LABEL_WARP=batched STP3_FUSED_ADAM=1 STP3_LAZY_BN_COUNTER=1 STP3_SE_MLP=1
