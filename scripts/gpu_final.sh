#!/bin/bash
# full GPU visit: all parity tests, lift timings, PMC passes, conv timings, eager bench (with cpu baseline),
# rocprof steady-state summary of the training step
TAG=${1:-final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 300 python scripts/time_lift.py > $OUT/time_lift.log 2>&1; grep -v amdgpu $OUT/time_lift.log
timeout 300 python scripts/time_conv.py 2>&1 | grep -v amdgpu > $OUT/time_conv.log; cat $OUT/time_conv.log
bash scripts/gpu_pmc.sh $TAG > $OUT/pmc_stdout.log 2>&1; head -12 $OUT/lift_pmc.json
timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json; cat $OUT/bench.json
bash scripts/gpu_prof.sh $TAG
