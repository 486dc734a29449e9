"""Per-dispatch listing of the LAST `window_ms` milliseconds of a rocprofv3 kernel trace (one training step), in launch
order: start offset (us), duration (us), grid, workgroup, kernel name (shortened).  Lets a kernel be attributed to the
layer that launched it.   python scripts/trace_last_step.py <kernel_trace.csv> <window_ms> > step_trace.txt"""
import re
import sys

import pandas as pd

path, window_ms = sys.argv[1], float(sys.argv[2])
df = pd.read_csv(path)
t_end = df['End_Timestamp'].max()
tail = df[df['Start_Timestamp'] >= t_end - window_ms * 1e6].sort_values('Start_Timestamp')
t0 = tail['Start_Timestamp'].min()
gx = [c for c in tail.columns if c.lower() in ('grid_size_x', 'grid_size')]
for _, r in tail.iterrows():
    name = re.sub(r'\(anonymous namespace\)::', '', str(r['Kernel_Name']))
    name = re.sub(r'^void ', '', name).split('(')[0][:90]
    grid = 'x'.join(str(int(r[c])) for c in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z') if c in tail.columns)
    wg = 'x'.join(str(int(r[c])) for c in ('Workgroup_Size_X', 'Workgroup_Size_Y', 'Workgroup_Size_Z') if c in tail.columns)
    print(f"{(r['Start_Timestamp'] - t0) / 1e3:10.1f} {(r['End_Timestamp'] - r['Start_Timestamp']) / 1e3:9.1f} {grid:>18} {wg:>9}  {name}")
