#!/bin/bash
# round 6, visit t: vectorised split-K batch reduce + the top-k selection kernel's serial chain: tests, then the same-box A/B
out=gpurun_out/r06t; mkdir -p $out
timeout 1500 python -m pytest tests/test_labels_gpu.py tests/test_conv_gpu.py tests/test_fused_ops_gpu.py tests/test_direct_bucket_grads_gpu.py tests/test_step_parity_gpu.py tests/test_train_parity_gpu.py tests/test_graph_step_gpu.py -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-200
bash scripts/gpu_r06e.sh
grep -E "topk_select|wgrad_reduce_batch|prep_weights" gpurun_out/r06e/steady_kernels.txt | cut -c1-150
