#!/bin/bash
# HBM traffic of the lift kernels from PMC counters, one counter per pass (TCC has 4 slots: FETCH_SIZE needs 3,
# WRITE_SIZE 2).   bash scripts/gpu_pmc.sh <tag>
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$TAG -o f -- python scripts/pmc_lift.py > $OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$TAG -o w -- python scripts/pmc_lift.py > $OUT/pmc_w.log 2>&1
python scripts/agg_pmc.py /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG > $OUT/lift_pmc.json 2> $OUT/agg_pmc.err
cat $OUT/lift_pmc.json; tail -3 $OUT/agg_pmc.err
rm -rf /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG
