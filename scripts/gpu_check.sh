#!/bin/bash
# One GPU-box visit: parity tests, lift kernel timings, bench line, rocprof kernel stats.
# Usage (from repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python scripts/time_lift.py > $OUT/time_lift.log 2>&1; cat $OUT/time_lift.log
timeout 600 python bench.py --steps 6 --warmup 3 > $OUT/bench.log 2>&1; tail -3 $OUT/bench.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1
tail -2 $OUT/prof.log
KT=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
[ -n "$KT" ] && python scripts/agg_trace.py $KT 0.55 70 > $OUT/steady_kernels.txt 2>&1 && head -40 $OUT/steady_kernels.txt
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
# traces are large; keep only the summaries
find $OUT/prof -name '*kernel_trace.csv' -delete
