#!/bin/bash
# Round-5 GPU visit.   bash scripts/gpu_r05.sh <tag> <what ...>
#   what: lifttests  voxel-pool GPU tests          liftab    time_lift.py + rocprofv3 kernel stats, r04 library vs this tree
#         liftpmc   HBM traffic counters of the pool kernels
#         suite      the whole GPU suite           parity    step / train parity tests with reports
#         quickbench bench.py without the CPU leg  bench     bench.py default flags + kernel-trace profile
#         convpmc    MFMA / wait / LDS counters of the convolution kernels per launched shape
#         steppmc    counters of every kernel of the (eager) step, per kernel name
out=gpurun_out/${1:-r05}; mkdir -p $out; shift
export STP3_PARITY_REPORT=$out/parity.json STP3_PARITY_REPORT_STEP=$out/parity_step.json STP3_IOU_REPORT=$out/iou.json TMPDIR=/tmp
has() { for w in "$@"; do for a in "${WHAT[@]}"; do [ "$a" == "$w" ] && return 0; done; done; return 1; }
WHAT=("$@")
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
if has lifttests; then
  timeout 900 python -m pytest tests/test_lift_gpu.py tests/test_lift_stress_gpu.py tests/test_voxsum_gpu.py -m gpu -q -x -p no:cacheprovider > $out/pytest_lift.log 2>&1
  echo "lift tests rc=$?"; tail -5 $out/pytest_lift.log | cut -c1-300
fi
if has liftab; then
  R04=$PWD/st-p3_amd/exp/libstp3hip_r04.so
  for v in r04 new; do
    [ $v == r04 ] && export EXP_LIB=$R04 || unset EXP_LIB
    timeout 200 python scripts/time_lift.py 4 > $out/time_lift_$v.log 2>&1; echo "--- $v library"; grep -E "plan build|bf16 BEV" $out/time_lift_$v.log | cut -c1-200
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lift_$v -o lift -- python scripts/time_lift.py 4 > $out/liftprof_$v.log 2>&1
    find /tmp/prof_lift_$v -name '*kernel_stats.csv' -exec cp {} $out/lift_kernel_stats_$v.csv \;
    python - <<PY
import csv
for r in csv.DictReader(open('$out/lift_kernel_stats_$v.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '')
    if 'lift_' in n or 'plan_' in n: print('   %-60s calls %4s avg %9.1f ns  min %s' % (n[:60], r['Calls'], float(r['AverageNs']), r['MinNs']))
PY
    rm -rf /tmp/prof_lift_$v
  done
  unset EXP_LIB
fi
if has liftpmc; then bash scripts/gpu_pmc_lift.sh $(basename $out) > $out/liftpmc.log 2>&1; tail -5 $out/liftpmc.log | cut -c1-300; fi
if has parity; then
  timeout 1500 python -m pytest tests/test_step_parity_gpu.py tests/test_train_parity_gpu.py tests/test_iou_gpu.py -m gpu -q -s -p no:cacheprovider > $out/pytest_parity.log 2>&1
  echo "parity tests rc=$?" | tee -a $out/pytest_parity.log; grep -E "passed|failed|^E  " $out/pytest_parity.log | cut -c1-400 | tail -20
fi
if has suite; then
  timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider $(has parity && echo "--deselect tests/test_step_parity_gpu.py --deselect tests/test_train_parity_gpu.py --deselect tests/test_iou_gpu.py") > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
  tail -8 $out/pytest.log | cut -c1-300
fi
if has convpmc; then bash scripts/gpu_pmc_conv.sh $(basename $out) > $out/convpmc.log 2>&1; tail -3 $out/convpmc.log | cut -c1-300; fi
if has steppmc; then bash scripts/gpu_pmc_step.sh $(basename $out) 2>&1 | tail -50 | cut -c1-160; fi
if has quickbench; then
  timeout 600 python bench.py --no-cpu-baseline > $out/bench_quick.json 2> $out/bench_quick.err; tail -3 $out/bench_quick.err
  python - <<PY
import json
b=json.load(open('$out/bench_quick.json'))
print('ms_per_step', b['ms_per_step'], 'samples/s', b['value'], 'host', b['host_enqueue_ms_per_step'], 'lift frac', b['roofline']['frac'], 'bwd', b['roofline']['backward'].get('frac'), b['kernel_ms'])
for k,v in b['roofline_families'].items():
    if isinstance(v,dict): print(' ', k, v.get('ms_per_step'), v.get('frac'), v.get('calls_per_step'))
PY
fi
if has bench || has profile; then
  if has bench; then timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cut -c1-1500 $out/bench.json; fi
  STEPS=4
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05 -o bench -- python scripts/bench_ab.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} > $out/prof.log 2>&1
  grep '^{' $out/prof.log | tail -1 > $out/bench_profiled.json
  KT=$(find /tmp/prof_r05 -name '*kernel_trace.csv' | head -1)
  MS=$(python -c "import json;print(json.load(open('$out/bench_profiled.json'))['ms_per_step'])" 2>/dev/null || echo 60)
  if [ -n "$KT" ]; then
    python scripts/agg_trace.py $KT $(python -c "print($MS*($STEPS-1))") 90 > $out/steady_kernels.txt 2>&1
    python scripts/trace_last_step.py $KT $MS > $out/step_trace.txt 2>&1
    python scripts/trace_overlap.py $KT $(python -c "print($MS*($STEPS-1))") > $out/trace_overlap.txt 2>&1; cat $out/trace_overlap.txt
  fi
  find /tmp/prof_r05 -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
  rm -rf /tmp/prof_r05
  head -40 $out/steady_kernels.txt | cut -c1-180
fi
du -sh gpurun_out
