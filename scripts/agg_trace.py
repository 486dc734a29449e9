"""Aggregate a rocprofv3 kernel_trace CSV over the LAST fraction of dispatches (steady state)."""
import sys
import pandas as pd
path, frac = sys.argv[1], float(sys.argv[2])
df = pd.read_csv(path)
df = df.sort_values('Start_Timestamp')
n = len(df)
tail = df.iloc[int(n * (1 - frac)):]
tail = tail.assign(dur=tail['End_Timestamp'] - tail['Start_Timestamp'])
span = (tail['End_Timestamp'].max() - tail['Start_Timestamp'].min()) / 1e6
g = tail.groupby('Kernel_Name')['dur'].agg(['count', 'sum', 'mean']).sort_values('sum', ascending=False)
tot = g['sum'].sum()
print(f'dispatches total {n}, tail {len(tail)}, tail wall {span:.1f} ms, tail kernel-time sum {tot/1e6:.1f} ms')
for name, r in g.head(int(sys.argv[3]) if len(sys.argv) > 3 else 45).iterrows():
    print(f"{r['sum']/1e6:10.2f} ms {100*r['sum']/tot:5.1f}% n={int(r['count']):5d} avg={r['mean']/1e3:9.1f} us  {name[:150]}")
