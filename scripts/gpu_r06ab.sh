#!/bin/bash
# round 6, visit ab (experiment): the depthwise forward-with-statistics kernel's pixel groups in one contiguous chunk per XCD
# (an EXPERIMENT build, not in the tree: launch_fwd_stats() of stp3_dwconv.hip read STP3_DW_XCD: 0 = grid-stride over all groups, 1 = chunks for
#  the 5 x 5 layers, 2 = chunks for every layer)
out=gpurun_out/r06ab; mkdir -p $out
for v in 0 1 2 0 1 2; do
  STP3_DW_XCD=$v timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dw xcd chunks $v:', d['ms_per_step'])" | tee -a $out/dw_xcd.txt
done
