#!/usr/bin/env python3
"""Timeline (start, duration, gap to previous kernel) of the last batch step in a rocprofv3 kernel trace."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_qbin" in r["Kernel_Name"] or "k_pretransform" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
tot_gap = 0
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    gx, gy = r["Grid_Size_X"], r["Grid_Size_Y"]
    tot_gap += max(0, s - prev_end)
    print(f"{n:24s} start {(s-t0)/1e3:9.1f} us dur {(e-s)/1e3:8.1f} gap {(s-prev_end)/1e3:7.1f} grid {gx}x{gy}")
    prev_end = e
print("total", (prev_end - t0) / 1e3, "us; gaps", tot_gap / 1e3, "us")
