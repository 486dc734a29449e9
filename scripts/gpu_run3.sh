#!/bin/bash
OUT=gpurun_out/r01i; mkdir -p $OUT
export STP3_MFMA_CONV=1
bash scripts/gpu_graph_probe.sh r01i temporal_fwd temporal
export MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_FWD_XDLOPS=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_GROUP_BWD_XDLOPS=0 MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_WRW_XDLOPS=0
export MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_FWD_XDLOPS=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS=0
echo "--- CK solvers disabled"
bash scripts/gpu_graph_probe.sh r01i_nock temporal full
STP3_MFMA_CONV=0 bash scripts/gpu_graph_probe.sh r01i_nock_nomfma temporal_fwd
