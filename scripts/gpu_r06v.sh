#!/bin/bash
# round 6, visit v: the decoder's written-out conv -> BatchNorm layers on the statistics epilogue: tests, then the same-box A/B
out=gpurun_out/r06v; mkdir -p $out
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_step_parity_gpu.py tests/test_train_parity_gpu.py tests/test_graph_step_gpu.py tests/test_graph_exchange_gpu.py tests/test_fused_ops_gpu.py -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-200
bash scripts/gpu_r06e.sh
grep -E "bn_stats_kernel|colsum|bn_reduce_partials" gpurun_out/r06e/steady_kernels.txt | cut -c1-150
