#!/bin/bash
# Counters of the convolution kernels (north_star: "MFMA utilisation on the convs"), per kernel AND launched shape.
#   bash scripts/gpu_pmc_conv.sh <tag>      -> gpurun_out/<tag>/conv_pmc.json, copy into profiles/
# Separate rocprofv3 passes per counter group (SQ: 8 slots, TCC: FETCH_SIZE / WRITE_SIZE alone), --kernel-trace only
# beside --pmc, while scripts/time_conv.py and scripts/time_pointwise.py run the model's dominant shapes.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_MFMA\|GRBM_GUI_ACTIVE" | sort -u > $OUT/counters_available.txt
pass() {  # name, counters...
  local name=$1; shift
  # (one rocprofv3 run per script: two processes under one run overwrite each other's output files)
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_conv_${TAG}_$name/conv -o c -- python scripts/time_conv.py > $OUT/pmc_conv_$name.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_conv_${TAG}_$name/pw -o c -- python scripts/time_pointwise.py >> $OUT/pmc_conv_$name.log 2>&1
}
MF=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES; do
  grep -qx "$c" $OUT/counters_available.txt && MF="$MF $c"
done
echo "mfma pass:$MF"
pass mfma $MF
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python scripts/agg_pmc_conv.py /tmp/pmc_conv_${TAG}_mfma /tmp/pmc_conv_${TAG}_wait /tmp/pmc_conv_${TAG}_tcc /tmp/pmc_conv_${TAG}_fetch /tmp/pmc_conv_${TAG}_write > $OUT/conv_pmc.json 2> $OUT/agg_pmc_conv.err
head -c 3000 $OUT/conv_pmc.json; tail -3 $OUT/agg_pmc_conv.err
grep -h "TF/s\|us" $OUT/pmc_conv_mfma.log | head -60 > $OUT/pmc_conv_shapes.txt
rm -rf /tmp/pmc_conv_${TAG}_*
