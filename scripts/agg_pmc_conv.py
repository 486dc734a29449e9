"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass over scripts/time_conv.py (see scripts/gpu_pmc_conv.sh).

    python scripts/agg_pmc_conv.py <dir>  -> JSON on stdout

For every kernel: mean of each collected counter per dispatch (summed over the counter's instances), the mean kernel
duration from the kernel trace, and -- when both are present -- SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024
SIMDs), i.e. the fraction of the chip's matrix-core issue slots that were busy (MI355X_MICROARCH.md: the counter is in
shader cycles, summed over SIMDs)."""
import glob
import json
import sys

import pandas as pd

CLOCK_HZ = 2.4e9
SIMDS = 256 * 4


def main():
    d = sys.argv[1]
    cc = pd.read_csv(glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0])
    per = cc.groupby(['Dispatch_Id', 'Kernel_Name', 'Counter_Name'])['Counter_Value'].sum().reset_index()
    tab = per.groupby(['Kernel_Name', 'Counter_Name'])['Counter_Value'].mean().unstack()
    counts = per.groupby('Kernel_Name')['Dispatch_Id'].nunique()
    dur = None
    traces = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if traces:
        kt = pd.read_csv(traces[0])
        kt['dur_us'] = (kt['End_Timestamp'] - kt['Start_Timestamp']) / 1e3
        dur = kt.groupby('Kernel_Name')['dur_us'].mean()
    out = {}
    for name, row in tab.iterrows():
        if not any(k in name for k in ('conv2d_', 'igemm', 'Cijk', 'ck::', 'dwconv')):
            continue
        ent = {'dispatches': int(counts[name]), **{k: float(v) for k, v in row.items() if v == v}}
        if dur is not None and name in dur.index:
            ent['avg_us'] = float(dur[name])
            busy = ent.get('SQ_VALU_MFMA_BUSY_CYCLES')
            if busy is not None and ent['avg_us'] > 0:
                ent['mfma_busy_frac'] = busy / (ent['avg_us'] * 1e-6 * CLOCK_HZ * SIMDS)
        out[name[:120]] = ent
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
