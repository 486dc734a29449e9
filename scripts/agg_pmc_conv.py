"""Counters of the convolution kernels per kernel AND launch geometry, from the rocprofv3 --pmc passes of
scripts/gpu_pmc_conv.sh over scripts/time_conv.py + scripts/time_pointwise.py.

    python scripts/agg_pmc_conv.py <dir> [<dir> ...]  -> JSON on stdout

One entry per (kernel name, grid size, workgroup size): the launched SHAPE of a kernel is identified by its grid (the
shape list of scripts/time_conv.py gives every shape its own grid).  Per entry: mean of each collected counter per
dispatch (summed over the counter's instances), the mean kernel duration from the kernel trace, and the derived
fractions -- SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) = share of the chip's matrix-core issue
slots that were busy (MI355X_MICROARCH.md: shader cycles summed over SIMDs), SQ_WAIT_ANY / SQ_WAVE_CYCLES, L2 hit rate,
HBM bytes (FETCH_SIZE x 2 x 32 B per the gfx950 correction measured in profiles/r04_lift_pmc.json, WRITE_SIZE x 64 B... the
raw counters are kept so that the correction can be re-applied)."""
import glob
import json
import sys

import pandas as pd

CLOCK_HZ = 2.4e9
SIMDS = 256 * 4


def key_cols(df):
    return ['Kernel_Name'] + [c for c in df.columns if c.startswith('Grid_Size') or c.startswith('Workgroup_Size')]


def main():
    out = {}
    for d in sys.argv[1:]:
        ccs = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))
        if not ccs:
            print('no counter_collection.csv under', d, file=sys.stderr)
            continue
        import os
        parts, durs = [], []
        for i, f in enumerate(ccs):                       # one file per profiled process: keep their dispatch ids apart
            part = pd.read_csv(f)
            part['Dispatch_Id'] = part['Dispatch_Id'] + i * 10_000_000
            parts.append(part)
            tr = f.replace('counter_collection', 'kernel_trace')
            if os.path.exists(tr):                        # the duration of EVERY dispatch, joined on its id (the trace has no grid columns)
                t = pd.read_csv(tr)
                if 'Dispatch_Id' in t.columns:
                    t = t.assign(Dispatch_Id=t['Dispatch_Id'] + i * 10_000_000, dur_us=(t['End_Timestamp'] - t['Start_Timestamp']) / 1e3)
                    durs.append(t[['Dispatch_Id', 'dur_us']])
        cc = pd.concat(parts, ignore_index=True)
        by_dispatch = pd.concat(durs, ignore_index=True).set_index('Dispatch_Id')['dur_us'] if durs else None
        keys = key_cols(cc)
        cc[keys[1:]] = cc[keys[1:]].fillna(-1)
        per = cc.groupby(['Dispatch_Id'] + keys + ['Counter_Name'])['Counter_Value'].sum().reset_index()
        tab = per.groupby(keys + ['Counter_Name'])['Counter_Value'].mean().unstack()
        counts = per.groupby(keys)['Dispatch_Id'].nunique()
        dur = None
        if by_dispatch is not None:
            one = per.drop_duplicates('Dispatch_Id').copy()
            one['dur_us'] = one['Dispatch_Id'].map(by_dispatch)
            dur = one.groupby(keys)['dur_us'].mean()
        for idx, row in tab.iterrows():
            idx = idx if isinstance(idx, tuple) else (idx,)
            name = idx[0]
            if not any(k in str(name) for k in ('conv2d_', 'igemm', 'pointwise', 'Cijk', 'dwconv', 'colsum')):
                continue
            label = str(name)[:110] + ''.join(f' | {c.lower()}={v}' for c, v in zip(keys[1:], idx[1:]))
            ent = out.setdefault(label, {})
            ent['dispatches'] = int(counts[idx if len(idx) > 1 else idx[0]])
            ent.update({k: float(v) for k, v in row.items() if v == v})
            if dur is not None:
                try:
                    ent.setdefault('avg_us', float(dur[idx if len(idx) > 1 else idx[0]]))   # (of the FIRST pass: the MFMA counters')
                except KeyError:
                    pass
    for ent in out.values():
        us = ent.get('avg_us')
        if us and 'SQ_VALU_MFMA_BUSY_CYCLES' in ent:
            ent['mfma_busy_frac'] = ent['SQ_VALU_MFMA_BUSY_CYCLES'] / (us * 1e-6 * CLOCK_HZ * SIMDS)
        if ent.get('SQ_WAVE_CYCLES') and 'SQ_WAIT_ANY' in ent:
            ent['wait_any_frac'] = ent['SQ_WAIT_ANY'] / ent['SQ_WAVE_CYCLES']
        if ent.get('TCC_REQ_sum'):
            ent['l2_hit_rate'] = ent.get('TCC_HIT_sum', 0.0) / ent['TCC_REQ_sum']
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
