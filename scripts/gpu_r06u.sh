#!/bin/bash
# round 6, visit u (experiment): the weight-gradient kernels' split count against the partial-sum traffic it causes
# (an EXPERIMENT build: wgrad_plan() of stp3_conv.hip read STP3_WGRAD_MIN_KSTEPS / STP3_WGRAD_ROUND_PCT for this visit; the
# switches are not in the tree -- the result, profiles/r06u_wgrad_split_policy.txt, kept the policy as it was)
out=gpurun_out/r06u; mkdir -p $out
for cfg in "8 100" "4 100" "6 100" "4 100" "8 100"; do
  set -- $cfg
  STP3_WGRAD_MIN_KSTEPS=$1 STP3_WGRAD_ROUND_PCT=$2 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('min_ksteps $1 round_pct $2:', d['ms_per_step'])" | tee -a $out/wgrad_splits.txt
done
