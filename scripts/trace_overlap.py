"""How much of a rocprofv3 kernel trace runs CONCURRENTLY: over the last `window_ms` of the trace, the sum of kernel
durations, the union of their busy intervals and the time at which two or more kernels were in flight, per queue.
    python scripts/trace_overlap.py <kernel_trace.csv> <window_ms>"""
import sys

import pandas as pd

path, window_ms = sys.argv[1], float(sys.argv[2])
df = pd.read_csv(path)
t_end = df['End_Timestamp'].max()
tail = df[df['Start_Timestamp'] >= t_end - window_ms * 1e6].sort_values('Start_Timestamp')
events = sorted([(s, 1) for s in tail['Start_Timestamp']] + [(e, -1) for e in tail['End_Timestamp']])
busy = multi = 0
depth, last = 0, events[0][0]
for t, d in events:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
total = (tail['End_Timestamp'] - tail['Start_Timestamp']).sum()
span = tail['End_Timestamp'].max() - tail['Start_Timestamp'].min()
print(f'window {span / 1e6:.2f} ms: kernel-time sum {total / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, '
      f'>= 2 kernels in flight {multi / 1e6:.2f} ms ({100 * multi / max(busy, 1):.1f} % of busy)')
qcol = next((c for c in ('Queue_Id', 'Stream_Id') if c in tail.columns), None)
if qcol:
    for q, g in tail.groupby(qcol):
        print(f'  {qcol} {q}: {len(g)} dispatches, {(g["End_Timestamp"] - g["Start_Timestamp"]).sum() / 1e6:.2f} ms')
