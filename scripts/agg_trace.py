"""Aggregate a rocprofv3 kernel_trace CSV over the LAST `window_ms` milliseconds (steady state):
   python scripts/agg_trace.py <kernel_trace.csv> <window_ms> [top_n]"""
import sys
import pandas as pd
path, window_ms = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
df = pd.read_csv(path)
t_end = df['End_Timestamp'].max()
tail = df[df['Start_Timestamp'] >= t_end - window_ms * 1e6]
tail = tail.assign(dur=tail['End_Timestamp'] - tail['Start_Timestamp'])
span = (tail['End_Timestamp'].max() - tail['Start_Timestamp'].min()) / 1e6
g = tail.groupby('Kernel_Name')['dur'].agg(['count', 'sum', 'mean']).sort_values('sum', ascending=False)
tot = g['sum'].sum()
print(f'dispatches total {len(df)}, in window {len(tail)}, window wall {span:.1f} ms, kernel-time sum {tot/1e6:.1f} ms '
      f'(GPU busy {100*tot/1e6/span:.0f}%)')
for name, r in g.head(top).iterrows():
    print(f"{r['sum']/1e6:9.3f} ms {100*r['sum']/tot:5.1f}% n={int(r['count']):5d} avg={r['mean']/1e3:9.1f} us  {name[:140]}")
