#!/bin/bash
OUT=gpurun_out/r01l; mkdir -p $OUT
timeout 120 python scripts/graph_memset_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/memset_probe.log
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q > $OUT/pytest_conv.log 2>&1; echo "conv pytest rc=$?"; tail -25 $OUT/pytest_conv.log
