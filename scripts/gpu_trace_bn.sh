#!/bin/bash
# Per-kernel durations of scripts/time_bn.py, shape by shape (kernel trace; the shapes run in the order of the script).
TAG=${1:-run}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_bn_$TAG -o p -- python scripts/time_bn.py > $OUT/trace_bn.log 2>&1
python - <<PY
import glob, re, pandas as pd
kt=pd.read_csv(glob.glob('/tmp/trace_bn_$TAG/**/*kernel_trace.csv', recursive=True)[0]).sort_values('Start_Timestamp')
kt['us']=(kt.End_Timestamp-kt.Start_Timestamp)/1e3
kt['k']=kt.Kernel_Name.str.extract(r'(bn_[a-z_]+|vectorized|elementwise)')[0]
kt=kt[kt.k.str.startswith('bn_', na=False)]
kt['grid']=kt.Grid_Size_X.astype(str)+'x'+kt.Grid_Size_Y.astype(str)+'x'+kt.Grid_Size_Z.astype(str)
# a new shape starts when the grid of bn_stats changes
g=kt.groupby(['grid','k'])['us'].agg(['count','median']).reset_index()
print(g.sort_values(['grid','k']).to_string())
PY
rm -rf /tmp/trace_bn_$TAG
