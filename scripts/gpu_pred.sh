#!/bin/bash
# Visit script of the prediction-stage work: its tests, the Prediction.yml bench line and,
# with "profile", a kernel trace of its step.   usage: gpurun -- bash scripts/gpu_pred.sh <tag> [tests] [bench] [profile] [planning]
out=gpurun_out/${1:-pred}; mkdir -p $out; shift
export TMPDIR=/tmp
has() { for a in "$@"; do [ "$a" == "$W" ] && return 0; done; return 1; }
for W in "$@"; do
  case $W in
    tests)
      timeout 900 python -m pytest tests/test_dwconv_gpu.py tests/test_prediction_gpu.py ${EXTRA_TESTS:-} -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1
      echo "tests rc=$?"; tail -4 $out/pytest.log | cut -c1-300 ;;
    bench)
      timeout 600 python bench.py --workload prediction --no-cpu-baseline --no-roofline --steps ${STEPS:-20} --warmup 3 > $out/bench_prediction.json 2> $out/bench_prediction.err
      tail -2 $out/bench_prediction.err; python -c "
import json; b=json.load(open('$out/bench_prediction.json')); print('prediction ms/step', b['ms_per_step'], 'samples/s', b['value'], b['config'].get('launch'), 'host', b.get('host_enqueue_ms_per_step'))" ;;
    planning)
      timeout 600 python bench.py --workload planning --no-cpu-baseline --no-roofline --steps ${STEPS:-20} --warmup 3 > $out/bench_planning.json 2> $out/bench_planning.err
      tail -2 $out/bench_planning.err; python -c "
import json; b=json.load(open('$out/bench_planning.json')); print('planning ms/step', b['ms_per_step'], 'samples/s', b['value'], b['config'].get('launch'), 'host', b.get('host_enqueue_ms_per_step'))" ;;
    profile)
      STEPS=4
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pred -o bench -- python bench.py --workload prediction --steps $STEPS --warmup 3 --no-cpu-baseline --no-roofline > $out/prof.log 2>&1
      grep '^{' $out/prof.log | tail -1 > $out/bench_profiled.json
      KT=$(find /tmp/prof_pred -name '*kernel_trace.csv' | head -1)
      MS=$(python -c "import json;print(json.load(open('$out/bench_profiled.json'))['ms_per_step'])" 2>/dev/null || echo 150)
      [ -n "$KT" ] && python scripts/agg_trace.py $KT $(python -c "print($MS*($STEPS-1))") 60 > $out/steady_kernels.txt 2>&1
      rm -rf /tmp/prof_pred; head -45 $out/steady_kernels.txt | cut -c1-170 ;;
    sources)
      timeout 360 python scripts/small_kernel_sources.py --workload ${WL:-prediction} --by-time --top 70 > $out/sources_${WL:-prediction}.txt 2> $out/sources.err
      cut -c1-230 $out/sources_${WL:-prediction}.txt | head -50 ;;
  esac
done
