#!/bin/bash
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for st in "$@"; do
  timeout 300 python scripts/graph_probe.py $st > $OUT/probe_$st.log 2>&1
  echo "== $st rc=$?"; grep -v "MIOpen(HIP)\|amdgpu.ids" $OUT/probe_$st.log | tail -6
done
