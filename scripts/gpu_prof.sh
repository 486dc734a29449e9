#!/bin/bash
# rocprofv3 kernel-trace of bench.py -> small steady-state summary only (traces are deleted:
# gpurun copies back at most 64 MiB).   bash scripts/gpu_prof.sh <tag> [bench args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/prof.log 2>&1
grep '^{' $OUT/prof.log | tail -1 > $OUT/bench_profiled.json
KT=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
MS=$(python -c "import json;print(json.load(open('$OUT/bench_profiled.json'))['ms_per_step']*($STEPS-1))" 2>/dev/null || echo 300)
[ -n "$KT" ] && python scripts/agg_trace.py $KT $MS 80 > $OUT/steady_kernels.txt 2>&1
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf /tmp/prof_$TAG
cat $OUT/bench_profiled.json; head -45 $OUT/steady_kernels.txt | cut -c1-200; du -sh gpurun_out
