#!/bin/bash
# round 6, visit c: the N > 1 step captured with its RCCL collectives (one rank, forced exchange path): which process-group settings survive
out=gpurun_out/r06c; mkdir -p $out
run() { # label, env...
  label=$1; shift
  for i in 1 2 3; do
    env "$@" timeout 300 python -m pytest tests/test_graph_exchange_gpu.py -q -x -s -p no:cacheprovider > $out/pytest_${label}_$i.log 2>&1; rc=$?
    echo "$label run $i rc=$rc $(grep -o 'hipErrorCapturedEvent' $out/pytest_${label}_$i.log | head -1) $(grep -E 'passed|failed' $out/pytest_${label}_$i.log | tail -1)"
  done
}
run base A=1
run nocache STP3_TEST_PG_ENV=TORCH_NCCL_CUDA_EVENT_CACHE=0
run drain STP3_GRAPH_DRAIN_S=0.5
run nocache_drain STP3_TEST_PG_ENV=TORCH_NCCL_CUDA_EVENT_CACHE=0 STP3_GRAPH_DRAIN_S=0.5
run noasync STP3_TEST_PG_ENV=TORCH_NCCL_ASYNC_ERROR_HANDLING=0,TORCH_NCCL_ENABLE_MONITORING=0
