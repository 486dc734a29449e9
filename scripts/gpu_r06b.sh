#!/bin/bash
# round 6, visit b: deferred weight-gradient sums (tests + bench), cold-operand probe of the BatchNorm kernels
out=gpurun_out/r06b; mkdir -p $out
timeout 900 python -m pytest tests/test_direct_bucket_grads_gpu.py tests/test_graph_step_gpu.py tests/test_conv_gpu.py tests/test_fused_ops_gpu.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -15 $out/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cut -c1-400 $out/bench.json
timeout 300 python scripts/time_bn_cold.py > $out/time_bn_cold.txt 2>&1; cat $out/time_bn_cold.txt
