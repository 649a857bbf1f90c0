#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel stats + the timeline of the last batch step."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    n = r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(n, [0, 0.0, 0.0, r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size']])
    a[0] += 1; a[1] += dur; a[2] = max(a[2], dur)
tot = sum(a[1] for a in agg.values())
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:28s} calls {a[0]:5d} total {a[1]/1e3:9.3f} ms avg {a[1]/a[0]:9.1f} us max {a[2]:9.1f} us  {100*a[1]/tot:5.1f}%  vgpr {a[3]} sgpr {a[4]} lds {a[5]}")
if len(sys.argv) > 2:
    sel = [r for r in rows if 'k_knn' in r['Kernel_Name']]
    t0 = int(sel[0]['Start_Timestamp'])
    for r in sel[-int(sys.argv[2]):]:
        n = r['Kernel_Name'].split('<')[0].split('::')[-1]
        print(f"  {n:12s} t={(int(r['Start_Timestamp'])-t0)/1e6:9.3f} ms dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us")
