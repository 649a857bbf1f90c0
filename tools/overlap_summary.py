#!/usr/bin/env python3
"""Host-input pipeline evidence from a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py: how much of the
host-to-device copy time of the uploads lies inside the time the GPU spends in kernels (copy / compute overlap)."""
import csv, glob, sys
d = sys.argv[1]
kern = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
cop = []
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'HOST_TO_DEVICE' in r.get('Direction', '').upper() or 'H2D' in r.get('Direction', '').upper():
            cop.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
kern.sort()
# merge kernel intervals
merged = []
for a, b in kern:
    if merged and a <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], b)
    else:
        merged.append([a, b])
# (the trace carries no sizes: the scans, 1.2 MB each, are the copies that take 15 us or more; descriptors take 1-3 us)
big = [c for c in cop if c[1] - c[0] >= 15_000]
def inside(a, b):
    t = 0
    for x, y in merged:
        if y <= a: continue
        if x >= b: break
        t += min(b, y) - max(a, x)
    return t
tot = sum(b - a for a, b in big)
ov = sum(inside(a, b) for a, b in big)
print(f"host-to-device copies of scan size (>= 15 us): {len(big)} of {len(cop)}, {tot / 1e6:.2f} ms of copy time, "
      f"{ov / 1e6:.2f} ms of it ({100.0 * ov / max(tot, 1):.1f} %) while kernels were running; "
      f"kernel-busy time {sum(b - a for a, b in merged) / 1e6:.2f} ms over a span of {(merged[-1][1] - merged[0][0]) / 1e6:.2f} ms")
