"""Per-kernel averages of PMC counters from rocprofv3 counter_collection CSVs (one directory per pass).
   python scripts/agg_pmc.py <pass dir> [<pass dir> ...]  -> JSON on stdout
For every kernel of interest: mean per dispatch of each counter (summed over the counter's instances), mean duration
from the kernel trace of the first pass.  FETCH_SIZE / WRITE_SIZE are in KiB; the calibration stream (torch.add over
256 MiB, known bytes) gives the correction factors for this access width on gfx950 (MI355X_MICROARCH.md section HBM:
FETCH_SIZE reads 1/2 of a wide coalesced stream), which are applied to give hbm_read_bytes / hbm_write_bytes."""
import glob, json, os, subprocess, sys
import pandas as pd

CAL_BYTES = (64 << 20) * 4
KEYS = ('lift_', 'plan_', 'depth_softmax', 'grad_import', 'transpose_kernel', 'voxel_index', 'add')


def shorten(name):
    if 'at::native' in name:
        return 'calibration_add' if ('add' in name.lower() and 'vectorized' in name) else 'aten:' + name[:50]
    return name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60]


def main():
    out, durs = {}, {}
    for d in sys.argv[1:]:
        files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
        if not files:
            continue
        df = pd.read_csv(files[0])
        per = df.groupby(['Dispatch_Id', 'Kernel_Name', 'Counter_Name'])['Counter_Value'].sum().reset_index()
        tab = per.groupby(['Kernel_Name', 'Counter_Name'])['Counter_Value'].mean().unstack()
        for name, row in tab.iterrows():
            if not any(k in name for k in KEYS):
                continue
            short = shorten(name)
            ent = out.setdefault(short, {})
            ent.update({k: float(v) for k, v in row.items() if v == v})
        tr = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
        if tr and not durs:
            kt = pd.read_csv(tr[0])
            kt['us'] = (kt['End_Timestamp'] - kt['Start_Timestamp']) / 1e3
            for name, g in kt.groupby('Kernel_Name'):
                if any(k in name for k in KEYS):
                    short = shorten(name)
                    durs[short] = (float(g['us'].mean()), int(len(g)))
    for k, (us, n) in durs.items():
        out.setdefault(k, {}).update({'avg_us_under_profiler': us, 'dispatches': n})
    cal = out.get('calibration_add', {})
    fc = CAL_BYTES / (cal['FETCH_SIZE'] * 1024.0) if cal.get('FETCH_SIZE') else None
    wc = CAL_BYTES / (cal['WRITE_SIZE'] * 1024.0) if cal.get('WRITE_SIZE') else None
    for k, ent in out.items():
        if 'FETCH_SIZE' in ent:
            ent['hbm_read_bytes'] = ent['FETCH_SIZE'] * 1024.0 * (fc or 1.0)
        if 'WRITE_SIZE' in ent:
            ent['hbm_write_bytes'] = ent['WRITE_SIZE'] * 1024.0 * (wc or 1.0)
        if ent.get('TCC_HIT_sum') is not None and ent.get('TCC_MISS_sum') is not None:
            ent['l2_hit_rate'] = ent['TCC_HIT_sum'] / max(ent['TCC_HIT_sum'] + ent['TCC_MISS_sum'], 1.0)
    try:
        commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], text=True).strip()
    except Exception:
        commit = os.environ.get('STP3_COMMIT')              # (the GPU box has no .git: the caller passes the commit of the snapshot)
    res = {'frames_per_launch': 12, 'commit': commit, 'calibration': {'true_bytes_each_way': CAL_BYTES, 'fetch_correction': fc,
                                                                       'write_correction': wc}}
    res.update(out)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
