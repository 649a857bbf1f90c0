#!/usr/bin/env python3
"""Every kernel (copy / fill kernels of the runtime included) of ONE scan of the facade in a rocprofv3 kernel trace: the launches
between the N-th last and the (N-1)-th last k_filter_points (a scan's input stage opens with it).  scan_timeline.py DIR [N=3]"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marks = [i for i, r in enumerate(rows) if "k_filter_points" in r["Kernel_Name"]]
a, b = marks[-back], marks[-back + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{n:32s} start {(s - t0) / 1e3:8.1f} us dur {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:7.1f} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev_end = e; busy += e - s
print(f"{b - a} launches, span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
