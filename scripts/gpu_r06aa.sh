#!/bin/bash
# round 6, visit aa (experiment): register blocking of the 5 x 5 depthwise forward-with-statistics kernel
# (an EXPERIMENT build, not in the tree: launch_fwd_stats() of stp3_dwconv.hip read STP3_DW5_VARIANT: 0 = 4 outputs per thread (172 VGPRs, 2 waves
#  per SIMD), 1 = 2 outputs / 3 waves (148), 2 = 2 outputs / 4 waves (128 + 48 B scratch), 3 = 4 outputs / 3 waves (168 + 32 B))
out=gpurun_out/r06aa; mkdir -p $out
for v in 0 4 5 0 4 5; do
  STP3_DW5_VARIANT=$v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dw5 variant $v:', d['ms_per_step'])" | tee -a $out/dw5_variants.txt
done
