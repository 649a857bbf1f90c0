#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/knn_traffic*.json: bytes over the
memory fabric per k_knn_grid launch.  Counter unit is KiB; on gfx950 FETCH_SIZE tallies every 128-byte line
request at 64 bytes, for 16-byte gathers exactly as for streaming reads (calibrated with
tools/micro/fetch_calib.hip: profiles/r02_fetch_size_calibration.txt), so the read side is doubled; WRITE_SIZE
is exact.  Requests served by the Infinity Cache are counted too: this is fabric traffic, an upper bound of
the DRAM traffic.

  pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json N_SCAN N_MAP BATCH [KERNEL [WORKLOAD]]"""
import csv, glob, json, os, sys
fetch_dir, write_dir, out, n_scan, n_map, batch = sys.argv[1:7]
kernel = sys.argv[7] if len(sys.argv) > 7 else "k_knn_grid"
workload = sys.argv[8] if len(sys.argv) > 8 else "icp_scan_to_map"
def mean_counter(d, name):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)
f, nf = mean_counter(fetch_dir, "FETCH_SIZE")
w, nw = mean_counter(write_dir, "WRITE_SIZE")
res = dict(commit=os.environ.get("PGSLAM_COMMIT"), workload=workload, n_scan=int(n_scan), n_map=int(n_map), batch=int(batch), kernel=kernel, launches=nf,
           fetch_size_kib_mean=f, write_size_kib_mean=w, fetch_correction=2.0,
           hbm_bytes_per_launch=(2.0 * f + w) * 1024.0, hbm_bytes_per_launch_uncorrected=(f + w) * 1024.0,
           note="fabric bytes (Infinity-Cache hits included); FETCH_SIZE x 2: every 128-byte line request is tallied at 64 bytes, "
                "gathers included (profiles/r02_fetch_size_calibration.txt)")
json.dump(res, open(out, "w"), indent=1)
print(res)
