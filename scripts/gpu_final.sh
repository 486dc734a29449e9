#!/bin/bash
# full GPU visit: all parity tests, PMC passes, eager bench (with cpu baseline), rocprof steady-state summary
TAG=${1:-final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 300 python scripts/time_lift.py > $OUT/time_lift.log 2>&1; grep -v amdgpu $OUT/time_lift.log
bash scripts/gpu_pmc.sh $TAG > $OUT/pmc_stdout.log 2>&1; head -12 $OUT/lift_pmc.json
bash scripts/gpu_prof.sh $TAG
