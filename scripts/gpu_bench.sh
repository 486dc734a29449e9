#!/bin/bash
# plain bench line (graph mode at N=1), no profiler.   bash scripts/gpu_bench.sh <tag> [bench args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python bench.py "$@" > $OUT/bench.log 2> $OUT/bench.err
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
cat $OUT/bench.json; grep -v "MIOpen\|amdgpu.ids" $OUT/bench.err | tail -15
