#!/bin/bash
# MFMA utilisation of the convolution kernels from PMC counters (north_star: "MFMA utilisation on the convs").
#   bash scripts/gpu_pmc_conv.sh <tag>      -> gpurun_out/<tag>/conv_mfma.json, copy into profiles/
# One SQ pass (8 slots on gfx950): matrix-core busy cycles against the kernel's busy cycles, per kernel, while
# scripts/time_conv.py runs the model's dominant convolution shapes.  Counter names differ between ROCm drops, so the
# script first lists what the box offers (rocprofv3 -L) and uses the MFMA / busy counters it finds.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_MFMA\|GRBM_GUI_ACTIVE" | sort -u > $OUT/counters_available.txt
cat $OUT/counters_available.txt
WANT=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  grep -qx "$c" $OUT/counters_available.txt && WANT="$WANT $c"
done
echo "collecting:$WANT"
[ -z "$WANT" ] && { echo "no MFMA counters offered by this rocprofv3"; exit 0; }
timeout 400 rocprofv3 --kernel-trace --pmc $WANT --output-format csv -d /tmp/pmc_conv_$TAG -o c -- python scripts/time_conv.py > $OUT/pmc_conv.log 2>&1
python scripts/agg_pmc_conv.py /tmp/pmc_conv_$TAG > $OUT/conv_mfma.json 2> $OUT/agg_pmc_conv.err
cat $OUT/conv_mfma.json | head -60; tail -3 $OUT/agg_pmc_conv.err
rm -rf /tmp/pmc_conv_$TAG
