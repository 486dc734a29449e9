"""Counters of every kernel of the training step, per kernel name: scripts/gpu_pmc_step.sh.

    python scripts/agg_pmc_step.py <steps profiled> <pass dir> [<pass dir> ...]  -> JSON on stdout

Only the dispatches of the LAST `steps` steps count (the profiled run also holds the set-up step and the warm-up): a kernel's
dispatches are ordered by id and the last  n / (steps + set-up + warm-up) * steps  are taken -- every step launches the same
kernels, so the split is exact for the steady state.  Per kernel name: dispatches and milliseconds per step (durations of the
FIRST pass, under the profiler), every counter summed per step, and derived -- HBM bytes (FETCH_SIZE x 1 KiB x 2.0, the gfx950
correction measured by the calibration stream of profiles/r05_lift_pmc.json; WRITE_SIZE x 1 KiB), wait fraction, L2 hit rate,
VALU instructions per HBM byte (the 'memory-bound kernel that is really instruction-bound' indicator of DESIGN section 5)."""
import glob
import json
import sys

import pandas as pd

FETCH_CORRECTION = 2.0
TOTAL_STEPS_EXTRA = 1 + 1 + 5          # bench.py --warmup 2: the set-up step, one more warm-up step, five host-enqueue steps


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    if name.startswith('at::native') or name.startswith('Cijk') or name.startswith('__amd'):
        return 'torch:' + name.split('<')[0][:40]
    return name.split('(')[0][:70]


def main():
    steps = int(sys.argv[1])
    out = {}
    first = True
    for d in sys.argv[2:]:
        ccs = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))
        if not ccs:
            print('no counter_collection.csv under', d, file=sys.stderr)
            continue
        cc = pd.read_csv(ccs[0])
        per = cc.groupby(['Dispatch_Id', 'Kernel_Name', 'Counter_Name'])['Counter_Value'].sum().reset_index()
        dur = None
        tr = ccs[0].replace('counter_collection', 'kernel_trace')
        try:
            t = pd.read_csv(tr)
            dur = t.assign(us=(t['End_Timestamp'] - t['Start_Timestamp']) / 1e3).set_index('Dispatch_Id')['us']
        except Exception as exc:                                    # noqa: BLE001
            print('no kernel trace for', d, exc, file=sys.stderr)
        per['short'] = per['Kernel_Name'].map(short)
        for name, g in per.groupby('short'):
            ids = sorted(g['Dispatch_Id'].unique())
            # steady state: the last `steps` of the (steps + extra) repetitions of this kernel
            n_total = len(ids)
            reps = steps + TOTAL_STEPS_EXTRA
            keep = ids[-max(1, round(n_total * steps / reps)):] if n_total >= reps else ids
            gk = g[g['Dispatch_Id'].isin(keep)]
            ent = out.setdefault(name, {})
            for cname, v in gk.groupby('Counter_Name')['Counter_Value'].sum().items():
                ent[cname] = float(v) / steps
            if first:
                ent['dispatches_per_step'] = len(keep) / steps
                if dur is not None:
                    ent['ms_per_step'] = float(dur.reindex(keep).sum()) / 1e3 / steps
        first = False
    for ent in out.values():
        if 'FETCH_SIZE' in ent:
            ent['hbm_read_bytes'] = ent['FETCH_SIZE'] * 1024.0 * FETCH_CORRECTION
        if 'WRITE_SIZE' in ent:
            ent['hbm_write_bytes'] = ent['WRITE_SIZE'] * 1024.0
        if ent.get('SQ_WAVE_CYCLES'):
            ent['wait_any_frac'] = ent.get('SQ_WAIT_ANY', 0.0) / ent['SQ_WAVE_CYCLES']
        if ent.get('TCC_REQ_sum'):
            ent['l2_hit_rate'] = ent.get('TCC_HIT_sum', 0.0) / ent['TCC_REQ_sum']
        b = ent.get('hbm_read_bytes', 0.0) + ent.get('hbm_write_bytes', 0.0)
        if b and 'SQ_INSTS_VALU' in ent:
            ent['valu_insts_per_byte'] = ent['SQ_INSTS_VALU'] * 64.0 / b          # lane-instructions per byte
    json.dump({'steps': steps, 'fetch_correction': FETCH_CORRECTION, 'kernels': out}, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
