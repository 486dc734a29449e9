#!/bin/bash
OUT=gpurun_out/r01h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q > $OUT/pytest_conv.log 2>&1; echo "conv pytest rc=$?"; tail -12 $OUT/pytest_conv.log
bash scripts/gpu_graph_probe.sh r01h temporal_fwd temporal_copy temporal_nodrop temporal_nopyr
