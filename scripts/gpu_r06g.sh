#!/bin/bash
# round 6, visit g: EXPERIMENT persistent igemm workgroups (.exp/libstp3hip_exp.so) against the in-tree library
out=gpurun_out/r06g; mkdir -p $out
EXP=$PWD/.exp/libstp3hip_exp.so
timeout 600 python -c "
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'st-p3_amd')
from stp3_amd import _lib; _lib.LIB_PATH='$EXP'
import pytest; sys.exit(pytest.main(['tests/test_conv_gpu.py','tests/test_fused_ops_gpu.py','-q','-x','-m','gpu','-p','no:cacheprovider']))" > $out/pytest_exp.log 2>&1; echo "exp tests rc=$?"; tail -3 $out/pytest_exp.log | cut -c1-200
timeout 600 python scripts/time_conv.py > $out/time_conv_base.txt 2>&1
EXP_LIB=$EXP timeout 600 python scripts/time_conv.py > $out/time_conv_exp.txt 2>&1
paste -d'|' <(cut -c1-110 $out/time_conv_base.txt) <(cut -c40-110 $out/time_conv_exp.txt) | head -60
STEPS=40 bash scripts/gpu_ab_lib.sh r06g_ab $EXP 2
