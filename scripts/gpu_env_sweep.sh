#!/bin/bash
# HIP runtime switches against the hipGraph step: a kernel of a few hundred threads takes 2.3-2.8 us when it is
# launched from the host (profiles/r05l_conv_pmc.json, eager micro-benchmarks) and 4.6-5.0 us as a node of the
# replayed graph (profiles/r05w_step_trace.txt) -- 960 of the step's 1891 kernels are that small.  This runs the
# headline bench under the runtime's graph / kernarg / fence switches and prints ms per step for each.
#   usage: gpurun -- bash scripts/gpu_env_sweep.sh <tag> ["VAR=VAL VAR=VAL" ...]   (no settings: the built-in list)
out=gpurun_out/${1:-envsweep}; mkdir -p $out; shift
export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" \
         "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=16" \
         "ROC_SYSTEM_SCOPE_SIGNAL=0"
fi
i=0
for setting in "$@"; do
  i=$((i + 1))
  log=$out/run_$i
  # shellcheck disable=SC2086
  env $setting timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps ${STEPS:-40} --warmup 5 \
      ${BENCH_ARGS:-} > $log.json 2> $log.err
  rc=$?
  python - "$setting" $rc $log.json <<'PY'
import json, sys
setting, rc, path = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    b = json.load(open(path))
    print(f"{setting or '(default)':42s} rc={rc} ms/step {b['ms_per_step']:.3f}  host {b.get('host_enqueue_ms_per_step')}  "
          f"launch {b['config'].get('launch')}")
except Exception as e:                                           # noqa: BLE001
    print(f"{setting or '(default)':42s} rc={rc} no bench line ({e})")
PY
done | tee $out/sweep.txt
