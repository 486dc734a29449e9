#!/bin/bash
# Same-session A/B of two builds of the library on the headline bench: the in-tree one and EXP (a path), alternating.
#   usage: gpurun -- bash scripts/gpu_ab_lib.sh <tag> <exp_lib> [reps] [pytest targets...]
out=gpurun_out/${1:-ab}; mkdir -p $out; exp=$2; reps=${3:-2}; shift 3
export TMPDIR=/tmp
if [ $# -gt 0 ]; then
  timeout 900 python -m pytest "$@" -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $out/pytest.log | cut -c1-200
fi
for r in $(seq 1 $reps); do
  for which in new prev; do
    if [ $which == prev ]; then export EXP_LIB=$exp; else unset EXP_LIB; fi
    timeout 300 python scripts/bench_ab.py --no-cpu-baseline --no-roofline --steps ${STEPS:-40} --warmup 5 ${BENCH_ARGS:-} > $out/${which}_$r.json 2> $out/${which}_$r.err
    python -c "
import json; b=json.load(open('$out/${which}_$r.json')); print('$which', $r, 'ms/step', b['ms_per_step'], b['config'].get('launch'), {k: v.get('ms_per_step') for k, v in (b.get('roofline_families') or {}).items() if isinstance(v, dict)})"
  done
done | tee $out/ab.txt
