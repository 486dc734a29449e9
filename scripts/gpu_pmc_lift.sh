#!/bin/bash
# Counters of the voxel-pool kernels, one rocprofv3 pass per counter group (TCC has 4 slots: FETCH_SIZE needs 3,
# WRITE_SIZE 2; SQ has 8).  No trace domains besides --kernel-trace are combined with --pmc.
#   bash scripts/gpu_pmc_lift.sh <tag>   -> gpurun_out/<tag>/lift_pmc.json (+ kernel durations)
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
pass() {  # name, counters...
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_${TAG}_$name -o p -- python scripts/pmc_lift.py > $OUT/pmc_$name.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
pass sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
python scripts/agg_pmc.py /tmp/pmc_${TAG}_fetch /tmp/pmc_${TAG}_write /tmp/pmc_${TAG}_tcc /tmp/pmc_${TAG}_sq /tmp/pmc_${TAG}_sq2 > $OUT/lift_pmc.json 2> $OUT/agg_pmc.err
cat $OUT/lift_pmc.json; tail -3 $OUT/agg_pmc.err; tail -2 $OUT/pmc_sq2.log
rm -rf /tmp/pmc_${TAG}_*
