#!/bin/bash
# round 6, visit z: the final tree on the N > 1 code path (one RCCL rank, bench.py --force-exchange) beside the plain step
out=gpurun_out/r06z; mkdir -p $out
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-workloads > $out/bench_plain.json 2> $out/plain.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-workloads --force-exchange > $out/bench_force_exchange_graph.json 2> $out/fx.err
for f in bench_plain bench_force_exchange_graph; do python -c "import json,sys; d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config'].get('launch'), d.get('collectives_per_step'))"; done
