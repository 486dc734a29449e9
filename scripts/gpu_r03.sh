#!/bin/bash
# Round-3 GPU visit.   bash scripts/gpu_r03.sh <tag> [quick]
#   new / changed operators first (fail fast) -> kernel micro-benchmarks -> parity (whole step, blocks) -> whole suite ->
#   bench line -> kernel-trace profile with the per-dispatch listing of the last step -> 2-rank launch attempt
out=gpurun_out/${1:-r03}; mkdir -p $out
export STP3_PARITY_REPORT=$out/parity.json STP3_PARITY_REPORT_STEP=$out/parity_step.json TMPDIR=/tmp
timeout 900 python -m pytest tests/test_datas_gpu.py tests/test_planning_gpu.py tests/test_conv_gpu.py tests/test_prediction_gpu.py tests/test_modules_gpu.py tests/test_fused_ops_gpu.py -m gpu -q -p no:cacheprovider > $out/pytest_new.log 2>&1
echo "new tests rc=$?" | tee -a $out/pytest_new.log; grep -E "passed|failed|Error|assert" $out/pytest_new.log | tail -8
if [ "$3" == "conv" ]; then timeout 300 python scripts/time_conv.py > $out/time_conv.log 2>&1; cut -c1-175 $out/time_conv.log; fi
if [ "$3" == "plan" ] || [ "$3" == "conv" ]; then timeout 300 python scripts/time_plan.py 4 > $out/time_plan.log 2>&1; tail -4 $out/time_plan.log
  timeout 300 python scripts/time_image.py 4 > $out/time_image.log 2>&1; tail -4 $out/time_image.log; fi
if [ "$3" == "lift" ]; then
  timeout 300 python scripts/time_lift.py 4 > $out/time_lift_c3.log 2>&1; tail -8 $out/time_lift_c3.log
  timeout 600 python scripts/time_lift.py 1 stress 5 > $out/time_lift_c5_b1.log 2>&1; tail -8 $out/time_lift_c5_b1.log
  timeout 600 python scripts/time_lift.py 4 stress 5 > $out/time_lift_c5_b4.log 2>&1; tail -8 $out/time_lift_c5_b4.log
fi
timeout 1500 python -m pytest tests/test_step_parity_gpu.py tests/test_train_parity_gpu.py -m gpu -q -s -p no:cacheprovider > $out/pytest_parity.log 2>&1
echo "parity tests rc=$?" | tee -a $out/pytest_parity.log; grep -E "passed|failed|\[step parity\]|^E  " $out/pytest_parity.log | cut -c1-400 | tail -30
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_step_parity_gpu.py --deselect tests/test_train_parity_gpu.py > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
  tail -8 $out/pytest.log
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
head -30 $out/steady_kernels.txt | cut -c1-180
if [ "$3" == "ranks" ]; then
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 180 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/two_ranks_one_gpu.log 2>&1; echo "2-rank launch on one GPU rc=$?"; grep -iE "error|duplicate|nccl|rccl" $out/two_ranks_one_gpu.log | head -5
fi
du -sh gpurun_out
