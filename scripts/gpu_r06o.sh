#!/bin/bash
out=gpurun_out/r06o; mkdir -p $out
timeout 900 python -m pytest tests/test_bench_exchange_gpu.py tests/test_graph_exchange_gpu.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log | cut -c1-300
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-exchange 2>/dev/null | grep '^{' | head -1 > $out/fx_graph_$i.json; python -c "
import json; d=json.load(open('$out/fx_graph_$i.json')); print('graph', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['collectives_per_step'])"; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-exchange --launch eager 2>/dev/null | grep '^{' | head -1 > $out/fx_eager.json; python -c "
import json; d=json.load(open('$out/fx_eager.json')); print('eager', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['collectives_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | head -1 > $out/plain.json; python -c "
import json; d=json.load(open('$out/plain.json')); print('plain', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
