"""Per-kernel averages of one PMC counter from rocprofv3 counter_collection CSVs.
   python scripts/agg_pmc.py <dir_with_FETCH_SIZE_run> <dir_with_WRITE_SIZE_run>  -> JSON on stdout
FETCH_SIZE / WRITE_SIZE are in KiB (guide section 7: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); the
calibration stream (torch.add over 256 MiB) gives the correction factor for this access width on gfx950
(MI355X_MICROARCH.md section HBM: FETCH_SIZE reads 1/2 of a wide coalesced stream)."""
import glob, json, sys
import pandas as pd

CAL_BYTES = (64 << 20) * 4


def load(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df['Counter_Name'] == counter]
    # a counter row may appear once per XCD/instance: sum per dispatch, then average per kernel
    per = df.groupby(['Dispatch_Id', 'Kernel_Name'])['Counter_Value'].sum().reset_index()
    return per.groupby('Kernel_Name')['Counter_Value'].agg(['mean', 'count'])


def pick(tab, key):
    rows = tab[tab.index.str.contains(key)]
    return None if rows.empty else float(rows['mean'].iloc[0]) * 1024.0


def pick_max(tab, key):
    rows = tab[tab.index.str.contains(key)]
    return None if rows.empty else float(rows['mean'].max()) * 1024.0


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
cal_f = pick_max(fetch, 'CUDAFunctorOnOther_add|AddFunctor|CUDAFunctor_add')
cal_w = pick_max(write, 'CUDAFunctorOnOther_add|AddFunctor|CUDAFunctor_add')
out = {'calibration': {'kernel': 'torch.add over 256 MiB float32', 'true_read_bytes': CAL_BYTES, 'true_write_bytes': CAL_BYTES,
                       'FETCH_SIZE_bytes': cal_f, 'WRITE_SIZE_bytes': cal_w,
                       'fetch_correction': None if not cal_f else CAL_BYTES / cal_f,
                       'write_correction': None if not cal_w else CAL_BYTES / cal_w}}
fc = out['calibration']['fetch_correction'] or 1.0
wc = out['calibration']['write_correction'] or 1.0
for name, key in (('lift_runs', 'lift_runs_kernel'), ('lift_gather', 'lift_gather_kernel'),
                  ('depth_softmax', 'depth_softmax_kernel'), ('bev_grad_accumulate', 'bev_grad_accumulate_kernel'),
                  ('lift_splat_bwd', 'lift_splat_bwd_kernel')):
    fb, wb = pick(fetch, key), pick(write, key)
    out[name] = {'FETCH_SIZE_bytes_raw': fb, 'WRITE_SIZE_bytes_raw': wb,
                 'hbm_read_bytes': None if fb is None else fb * fc, 'hbm_write_bytes': None if wb is None else wb * wc}
print(json.dumps(out, indent=1))
