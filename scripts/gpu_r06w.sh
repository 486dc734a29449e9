#!/bin/bash
# round 6, visit w (experiment): how many bytes of split-K partial sums stay in flight before a launch sums them
# (an EXPERIMENT build: ops._WgradArena summed its partial sums every STP3_WGRAD_ROLL_MB for this visit; the switch is not in
# the tree -- the result, profiles/r06w_wgrad_rolling_reduce.txt: one launch at the end of the pass stays)
out=gpurun_out/r06w; mkdir -p $out
for mb in 0 64 0 64 128 0; do
  STP3_WGRAD_ROLL_MB=$mb timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('roll_mb $mb:', d['ms_per_step'])" | tee -a $out/wgrad_roll.txt
done
