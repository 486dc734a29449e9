#!/bin/bash
# round 6, visit s: the squeeze-excite gate kernels with their loads in explicit batches: tests, then the same-box A/B of visit e
out=gpurun_out/r06s; mkdir -p $out
timeout 1500 python -m pytest tests/test_fused_ops_gpu.py tests/test_modules_gpu.py tests/test_step_parity_gpu.py tests/test_graph_step_gpu.py -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-200
bash scripts/gpu_r06e.sh
grep -E "se_mlp" gpurun_out/r06e/steady_kernels.txt | cut -c1-150
