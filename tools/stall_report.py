#!/usr/bin/env python3
"""Where do the 25-40 ms stalls of a cold slam_run pass come from?  Reads a rocprofv3 --kernel-trace --hip-runtime-trace CSV directory:
HIP API calls over 5 ms, kernels over 5 ms, and gaps over 5 ms on the kernel timeline with the kernels either side."""
import csv, glob, sys
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5e6
def load(pat):
    f = glob.glob(d + "/**/*" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
api = load("hip_api_trace.csv")
ker = load("kernel_trace.csv")
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(ker[0]["Start_Timestamp"]) if ker else 0
name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
print("HIP API calls over %.0f ms:" % (thr / 1e6))
slow = [r for r in api if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > thr]
for r in slow[:60]:
    print("  %-28s %8.2f ms at %9.2f ms" % (r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e6))
print("  (%d calls, %.1f ms in all)" % (len(slow), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in slow) / 1e6))
print("kernels over the threshold:")
for r in ker:
    if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > thr:
        print("  %-24s %8.2f ms at %9.2f ms queue %s" % (name(r), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("Queue_Id")))
print("gaps on the kernel timeline over the threshold:")
prev = None
n = tot = 0
for r in ker:
    if prev is not None:
        g = int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
        if g > thr:
            n += 1; tot += g
            if n <= 60:
                print("  %8.2f ms at %9.2f ms   after %-22s (queue %s)  before %-22s (queue %s)" % (g / 1e6, (int(prev["End_Timestamp"]) - t0) / 1e6, name(prev), prev.get("Queue_Id"), name(r), r.get("Queue_Id")))
    if prev is None or int(r["End_Timestamp"]) > int(prev["End_Timestamp"]):
        prev = r
print("  (%d gaps, %.1f ms in all; trace spans %.1f ms, %d kernels)" % (n, tot / 1e6, (int(ker[-1]["End_Timestamp"]) - t0) / 1e6 if ker else 0, len(ker)))
