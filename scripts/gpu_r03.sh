#!/bin/bash
# Round-3 GPU visit: new tests first (fail fast), the whole -m gpu suite, one bench line, a kernel-trace profile with the
# per-dispatch listing of the last step.   bash scripts/gpu_r03.sh <tag> [quick]
out=gpurun_out/${1:-r03}; mkdir -p $out
export STP3_PARITY_REPORT=$out/parity.json STP3_PARITY_REPORT_STEP=$out/parity_step.json TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fused_ops_gpu.py tests/test_step_parity_gpu.py -m gpu -q -s -p no:cacheprovider > $out/pytest_new.log 2>&1
echo "new tests rc=$?" | tee -a $out/pytest_new.log
grep -E "passed|failed|\[step parity\]|Error|assert" $out/pytest_new.log | tail -30
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --deselect tests/test_step_parity_gpu.py > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
  grep -v "^\[parity\]" $out/pytest.log | tail -15
fi
timeout 900 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
STEPS=4
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -o bench -- python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
grep '^{' $out/prof.log | tail -1 > $out/bench_profiled.json
KT=$(find /tmp/prof_r03 -name '*kernel_trace.csv' | head -1)
MS=$(python -c "import json;print(json.load(open('$out/bench_profiled.json'))['ms_per_step'])" 2>/dev/null || echo 60)
if [ -n "$KT" ]; then
  python scripts/agg_trace.py $KT $(python -c "print($MS*($STEPS-1))") 90 > $out/steady_kernels.txt 2>&1
  python scripts/trace_last_step.py $KT $MS > $out/step_trace.txt 2>&1
fi
find /tmp/prof_r03 -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf /tmp/prof_r03
head -40 $out/steady_kernels.txt | cut -c1-180; du -sh gpurun_out
