#!/bin/bash
# round 6, visit y (experiment): 128 x 64 tiles for the tiled convolutions whose 128 x 128 tiling fills less than the chip
# (an EXPERIMENT build: igemm_run() of stp3_conv.hip read STP3_IGEMM_NARROW_BELOW = workgroups of the wide tiling below which
# the narrow tile is taken; the tree now carries the winner, 512, as a constant)
out=gpurun_out/r06y; mkdir -p $out
for t in 0 256 0 256 512 0 1024; do
  STP3_IGEMM_NARROW_BELOW=$t timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('narrow_below $t:', d['ms_per_step'])" | tee -a $out/igemm_narrow.txt
done
