#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: sum / mean of every counter."""
import csv, glob, sys, collections
d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else None
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
        if want and want not in n:
            continue
        agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for n, cs in agg.items():
        for c, v in sorted(cs.items()):
            print(f"{n:20s} {c:32s} dispatches {len(v):4d} sum {sum(v):16.0f} mean {sum(v)/len(v):14.1f} max {max(v):14.0f}")

# per-dispatch dump (argv[3] == 'each')
if len(sys.argv) > 3 and sys.argv[3] == 'each':
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = collections.OrderedDict()
        for r in rows:
            n = r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
            if want and want not in n:
                continue
            per.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
        for k, v in per.items():
            print(k, ' '.join(f"{c}={x:.3g}" for c, x in sorted(v.items())))
