#!/bin/bash
# round 6, visit q: weights assembled from parameter views (ops.assembled_weight): tests, then the same-box A/B of visit e
out=gpurun_out/r06q; mkdir -p $out
timeout 1500 python -m pytest tests/test_fused_ops_gpu.py tests/test_modules_gpu.py tests/test_train_parity_gpu.py tests/test_step_parity_gpu.py tests/test_graph_step_gpu.py tests/test_graph_exchange_gpu.py tests/test_direct_bucket_grads_gpu.py tests/test_prediction_gpu.py -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-200
bash scripts/gpu_r06e.sh
