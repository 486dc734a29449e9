#!/bin/bash
# round 6, visit h: the prediction / planning legs as full bench lines (with the reference's CPU step), smoke()
out=gpurun_out/r06h; mkdir -p $out
timeout 300 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
for w in prediction planning; do
  timeout 1500 python bench.py --workload $w --steps 20 --warmup 5 > $out/bench_$w.json 2> $out/bench_$w.err; tail -2 $out/bench_$w.err | cut -c1-200
  python -c "
import json; d=json.load(open('$out/bench_$w.json')); print('$w', d['ms_per_step'], d['value'], d['config']['launch'], 'cpu', {k: d['cpu_baseline'].get(k) for k in ('value','kind','cores','step_s_median_of_3','error')})"
done
