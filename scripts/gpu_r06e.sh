#!/bin/bash
# round 6, visit e: same-box A/B of the round-5 tree (.ab_r05, commit 93f4b96) against HEAD, then a kernel trace of HEAD
# (.ab_r05 is git-ignored scratch that travels with gpurun:  mkdir .ab_r05 && git archive 93f4b96 | tar -x -C .ab_r05 &&
#  make -C .ab_r05/st-p3_amd/csrc -j8 && cp -r oracle/_ref .ab_r05/oracle/)
out=gpurun_out/r06e; mkdir -p $out
export TMPDIR=/tmp
for i in 1 2; do
  (cd .ab_r05 && timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r05 ', d['ms_per_step'], d['config']['launch'])") | tee -a $out/ab.txt
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEAD', d['ms_per_step'], d['config']['launch'])" | tee -a $out/ab.txt
done
STEPS=4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r06e -o bench -- python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
grep '^{' $out/prof.log | tail -1 > $out/bench_profiled.json
KT=$(find /tmp/prof_r06e -name '*kernel_trace.csv' | head -1)
MS=$(python -c "import json;print(json.load(open('$out/bench_profiled.json'))['ms_per_step']*($STEPS-1))" 2>/dev/null || echo 110)
[ -n "$KT" ] && python scripts/agg_trace.py $KT $MS 80 > $out/steady_kernels.txt 2>&1
MS1=$(python -c "import json;print(json.load(open('$out/bench_profiled.json'))['ms_per_step'])" 2>/dev/null || echo 38)
[ -n "$KT" ] && python scripts/trace_last_step.py $KT $MS1 > $out/step_trace.txt 2>&1
find /tmp/prof_r06e -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf /tmp/prof_r06e
head -3 $out/steady_kernels.txt | cut -c1-200; wc -l $out/step_trace.txt
