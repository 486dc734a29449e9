#!/bin/bash
# round 6, visit d: merged small reductions of the MBConv middle (tests + bench), the forced-exchange step under capture timed
out=gpurun_out/r06d; mkdir -p $out
timeout 1200 python -m pytest tests/test_fused_ops_gpu.py tests/test_dwconv_gpu.py tests/test_graph_step_gpu.py tests/test_graph_exchange_gpu.py tests/test_train_parity_gpu.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -8 $out/pytest.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cut -c1-300 $out/bench.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-exchange > $out/bench_force_exchange.json 2> $out/bench_fx.err; tail -3 $out/bench_fx.err; cut -c1-300 $out/bench_force_exchange.json; python -c "
import json; d=json.load(open('$out/bench_force_exchange.json')); print(d['config']['launch'], d['config']['parallelism'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['collectives_per_step'])"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --force-exchange --launch eager > $out/bench_force_exchange_eager.json 2>> $out/bench_fx.err; python -c "
import json; d=json.load(open('$out/bench_force_exchange_eager.json')); print(d['config']['launch'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['collectives_per_step'])"
