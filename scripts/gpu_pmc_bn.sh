#!/bin/bash
# SQ counters of the BatchNorm kernels over scripts/time_bn.py (is a streaming kernel waiting or issuing?)
TAG=${1:-run}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d /tmp/pmc_bn_$TAG -o p -- python scripts/time_bn.py > $OUT/pmc_bn.log 2>&1
python - <<PY
import glob, pandas as pd
d='/tmp/pmc_bn_$TAG'
cc=pd.read_csv(glob.glob(d+'/**/*counter_collection.csv', recursive=True)[0])
kt=pd.read_csv(glob.glob(d+'/**/*kernel_trace.csv', recursive=True)[0]); kt['us']=(kt.End_Timestamp-kt.Start_Timestamp)/1e3
dur=kt.set_index('Dispatch_Id')['us']
per=cc.groupby(['Dispatch_Id','Kernel_Name','Counter_Name'])['Counter_Value'].sum().unstack().reset_index()
per['us']=per.Dispatch_Id.map(dur)
per=per[per.Kernel_Name.str.contains('bn_apply_fwd|bn_bwd_reduce|bn_apply_bwd|bn_stats')]
per['k']=per.Kernel_Name.str.extract(r'(bn_[a-z_]+)_kernel')[0]
# the largest dispatch of each kernel type
for k,g in per.groupby('k'):
    r=g.sort_values('us').iloc[-1]
    simd_cycles=r.us*1e-6*2.4e9*1024
    print(f"{k:16s} {r.us:7.1f} us  waves {r.SQ_WAVES:8.0f}  VALU insts/wave {r.SQ_INSTS_VALU/r.SQ_WAVES:7.0f}  SALU/wave {r.SQ_INSTS_SALU/r.SQ_WAVES:6.0f}  valu_busy {4*r.SQ_ACTIVE_INST_VALU/simd_cycles:5.2f}  any_busy {4*r.SQ_ACTIVE_INST_ANY/simd_cycles:5.2f}  wave_occupancy {4*r.SQ_WAVE_CYCLES/simd_cycles:5.2f} waves/SIMD")
PY
rm -rf /tmp/pmc_bn_$TAG
