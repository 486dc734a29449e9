#!/bin/bash
# Where does the weight-gradient kernel wait?  FETCH_SIZE / TCC hit-miss / SQ wait counters of scripts/time_conv.py on ONE
# layer shape (default: the 3x3 64->64 decoder head at 200x200x12), one rocprofv3 pass per counter group.
#   bash scripts/gpu_pmc_wgrad.sh <tag> ["layer name filter"]   -> gpurun_out/<tag>/wgrad_pmc.json
TAG=${1:-run}; FILTER=${2:-decoder head}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
pass() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcw_${TAG}_$name -o p -- python scripts/time_conv.py "$FILTER" > $OUT/pmcw_$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
python - "$TAG" > $OUT/wgrad_pmc.json <<'PY'
import glob, json, sys
import pandas as pd
tag = sys.argv[1]
out = {}
for d in sorted(glob.glob(f'/tmp/pmcw_{tag}_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        df = pd.read_csv(f)
        per = df.groupby(['Dispatch_Id', 'Kernel_Name', 'Counter_Name'])['Counter_Value'].sum().reset_index()
        tab = per.groupby(['Kernel_Name', 'Counter_Name'])['Counter_Value'].mean().unstack()
        for name, row in tab.iterrows():
            if 'conv2d' not in name:
                continue
            short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:50]
            out.setdefault(short, {}).update({k: float(v) for k, v in row.items() if v == v})
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        kt = pd.read_csv(f)
        kt['us'] = (kt['End_Timestamp'] - kt['Start_Timestamp']) / 1e3
        for name, g in kt.groupby('Kernel_Name'):
            if 'conv2d' in name:
                short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:50]
                out.setdefault(short, {}).setdefault('avg_us', float(g['us'].mean()))
for k, e in out.items():
    if 'FETCH_SIZE' in e: e['fetch_MB_raw_KiB_units'] = e['FETCH_SIZE'] * 1024 / 1e6
    if 'WRITE_SIZE' in e: e['write_MB_raw_KiB_units'] = e['WRITE_SIZE'] * 1024 / 1e6
    if 'TCC_HIT_sum' in e and 'TCC_MISS_sum' in e: e['l2_hit_rate'] = e['TCC_HIT_sum'] / max(e['TCC_HIT_sum'] + e['TCC_MISS_sum'], 1.0)
    if 'SQ_WAIT_ANY' in e and 'SQ_WAVE_CYCLES' in e: e['wait_any_frac'] = e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']
    if 'SQ_WAIT_INST_LDS' in e and 'SQ_WAVE_CYCLES' in e: e['wait_lds_frac'] = e['SQ_WAIT_INST_LDS'] / e['SQ_WAVE_CYCLES']
print(json.dumps(out, indent=1))
PY
cat $OUT/wgrad_pmc.json
rm -rf /tmp/pmcw_${TAG}_*
