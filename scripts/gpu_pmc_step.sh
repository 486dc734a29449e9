#!/bin/bash
# Counters of EVERY kernel of the training step, per kernel name, from rocprofv3 --pmc passes over two eager steps of bench.py
# (the captured step replays the same kernels; counters need the individual dispatches).
#   bash scripts/gpu_pmc_step.sh <tag>   -> gpurun_out/<tag>/step_pmc.json (copy into profiles/)
# One pass per counter group, --kernel-trace only beside --pmc.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=2
pass() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_step_${TAG}_$name -o s -- python bench.py --launch eager --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline > $OUT/pmc_step_$name.log 2>&1
}
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python scripts/agg_pmc_step.py $STEPS /tmp/pmc_step_${TAG}_sq /tmp/pmc_step_${TAG}_sq2 /tmp/pmc_step_${TAG}_tcc /tmp/pmc_step_${TAG}_fetch /tmp/pmc_step_${TAG}_write > $OUT/step_pmc.json 2> $OUT/agg_pmc_step.err
python - <<PY
import json
d = json.load(open('$OUT/step_pmc.json'))
rows = sorted(d['kernels'].items(), key=lambda kv: -kv[1].get('ms_per_step', 0))[:45]
print('%-52s %6s %8s %8s %7s %6s %6s %6s' % ('kernel', 'n/step', 'ms/step', 'GB/step', 'TB/s', 'wait', 'l2hit', 'valu/B'))
for k, v in rows:
    gb = (v.get('hbm_read_bytes', 0) + v.get('hbm_write_bytes', 0)) / 1e9
    ms = v.get('ms_per_step', 0)
    print('%-52s %6.0f %8.3f %8.3f %7.2f %6.2f %6.2f %6.2f' % (k[:52], v.get('dispatches_per_step', 0), ms, gb, gb / ms if ms else 0,
          v.get('wait_any_frac', 0), v.get('l2_hit_rate', 0), v.get('valu_insts_per_byte', 0)))
PY
tail -3 $OUT/agg_pmc_step.err
rm -rf /tmp/pmc_step_${TAG}_*
