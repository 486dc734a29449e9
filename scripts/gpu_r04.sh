#!/bin/bash
# Round-4 GPU visit.   bash scripts/gpu_r04.sh <tag> <what ...>
#   what: parity  the new / changed parity tests first (fail fast), the noise probe of the whole step
#         suite   the whole GPU suite
#         bench   bench line (default flags) + kernel-trace profile with the per-dispatch listing of the last step
#         conv | pointwise | lift | plan   micro-benchmarks
out=gpurun_out/${1:-r04}; mkdir -p $out; shift
export STP3_PARITY_REPORT=$out/parity.json STP3_PARITY_REPORT_STEP=$out/parity_step.json STP3_IOU_REPORT=$out/iou.json TMPDIR=/tmp
has() { for w in "$@"; do for a in "${WHAT[@]}"; do [ "$a" == "$w" ] && return 0; done; done; return 1; }
WHAT=("$@")
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
if has parity; then
  timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_iou_gpu.py tests/test_planning_gpu.py tests/test_modules_gpu.py -m gpu -q -s -p no:cacheprovider --deselect tests/test_conv_gpu.py::test_conv2d_on_tensors_beyond_2g_elements > $out/pytest_new.log 2>&1
  echo "new tests rc=$?" | tee -a $out/pytest_new.log; grep -E "passed|failed|\[iou b4\]|^E  " $out/pytest_new.log | cut -c1-600 | tail -14
  timeout 1500 python -m pytest tests/test_step_parity_gpu.py tests/test_train_parity_gpu.py -m gpu -q -s -p no:cacheprovider > $out/pytest_parity.log 2>&1
  echo "parity tests rc=$?" | tee -a $out/pytest_parity.log; grep -E "passed|failed|\[step parity\]|^E  " $out/pytest_parity.log | cut -c1-400 | tail -30
  timeout 900 python scripts/step_noise_probe.py --draws 5 --out $out/step_noise_probe.json > $out/noise_probe.log 2>&1; echo "noise probe rc=$?"; grep -v Warning $out/noise_probe.log | cut -c1-300 | tail -12
fi
if has suite; then
  timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider $(has parity && echo "--deselect tests/test_step_parity_gpu.py --deselect tests/test_train_parity_gpu.py --deselect tests/test_iou_gpu.py") > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
  tail -8 $out/pytest.log
fi
if has conv; then timeout 300 python scripts/time_conv.py > $out/time_conv.log 2>&1; cut -c1-175 $out/time_conv.log; fi
if has pointwise; then timeout 300 python scripts/time_pointwise.py > $out/time_pointwise.log 2>&1; cut -c1-260 $out/time_pointwise.log; fi
if has plan; then timeout 300 python scripts/time_plan.py 4 > $out/time_plan.log 2>&1; tail -4 $out/time_plan.log; fi
if has lift; then timeout 300 python scripts/time_lift.py 4 > $out/time_lift_c3.log 2>&1; tail -8 $out/time_lift_c3.log; fi
if has perftests; then
  timeout 1200 python -m pytest tests/test_fused_ops_gpu.py tests/test_bnact_gpu.py tests/test_modules_gpu.py tests/test_train_parity_gpu.py tests/test_step_parity_gpu.py -m gpu -q -x -p no:cacheprovider > $out/pytest_perf.log 2>&1
  echo "perf-change tests rc=$?"; tail -4 $out/pytest_perf.log | cut -c1-300
fi
if has quickbench; then
  timeout 600 python bench.py --no-cpu-baseline > $out/bench_quick.json 2> $out/bench_quick.err; tail -3 $out/bench_quick.err
  python - <<PY
import json
b=json.load(open('$out/bench_quick.json'))
print('ms_per_step', b['ms_per_step'], 'samples/s', b['value'], 'lift frac', b['roofline']['frac'])
for k,v in b['roofline_families'].items():
    if isinstance(v,dict): print(' ', k, v.get('ms_per_step'), v.get('frac'), v.get('calls_per_step'))
PY
fi
if has bench || has profile; then
  if has bench; then timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cut -c1-1500 $out/bench.json; fi
  STEPS=4
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04 -o bench -- python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
  grep '^{' $out/prof.log | tail -1 > $out/bench_profiled.json
  KT=$(find /tmp/prof_r04 -name '*kernel_trace.csv' | head -1)
  MS=$(python -c "import json;print(json.load(open('$out/bench_profiled.json'))['ms_per_step'])" 2>/dev/null || echo 60)
  if [ -n "$KT" ]; then
    python scripts/agg_trace.py $KT $(python -c "print($MS*($STEPS-1))") 90 > $out/steady_kernels.txt 2>&1
    python scripts/trace_last_step.py $KT $MS > $out/step_trace.txt 2>&1
  fi
  find /tmp/prof_r04 -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
  rm -rf /tmp/prof_r04
  head -30 $out/steady_kernels.txt | cut -c1-180
fi
du -sh gpurun_out
