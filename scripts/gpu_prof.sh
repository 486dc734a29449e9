#!/bin/bash
# rocprofv3 kernel-trace of bench.py -> small steady-state summary only (traces are deleted:
# gpurun copies back at most 64 MiB).   bash scripts/gpu_prof.sh <tag> [bench args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/prof.log 2>&1
grep '^{' $OUT/prof.log | tail -1 > $OUT/bench_profiled.json
KT=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
[ -n "$KT" ] && python scripts/agg_trace.py $KT 0.55 90 > $OUT/steady_kernels.txt 2>&1
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf /tmp/prof_$TAG
cat $OUT/bench_profiled.json; head -50 $OUT/steady_kernels.txt; du -sh gpurun_out
