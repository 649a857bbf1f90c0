#!/bin/bash
# round 5: why did slam_100k's ICP phase take 3.0 s on the driver's box and 0.6 s in the builder's record?
# (a) standalone leg, 1 warm + 3 timed passes in one slam_run process; (b) the same slam_run under a parent that holds a GPU
# context (what the default bench line does: the leg is a child of a process that has run the headline); (c) the default line.
TAG=${1:-sd}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --steps 1 --warmup 0 --no-cpu-baseline > $OUT/a_standalone.json 2> $OUT/err.log
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
./tools/slam_run $SEQ --filters sensor --passes 1 > $OUT/a_cold_single_pass.json 2>> $OUT/err.log
python - > $OUT/b_under_gpu_parent.json 2>> $OUT/err.log <<PY
import subprocess, sys, torch
sys.path.insert(0, ".")
from pgslam_amd import icp
x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
c = icp.Context(0)
out = subprocess.run(["./tools/slam_run", "$SEQ", "--filters", "sensor", "--passes", "4"], capture_output=True, text=True)
print(out.stdout.strip().splitlines()[-1])
c.close()
out = subprocess.run(["./tools/slam_run", "$SEQ", "--filters", "sensor", "--passes", "1"], capture_output=True, text=True)
print(out.stdout.strip().splitlines()[-1])
PY
python bench.py --steps 20 --warmup 5 > $OUT/c_default_line.json 2>> $OUT/err.log
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    for ln in open(f).read().strip().splitlines():
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        s = d.get("slam", d)
        if "workloads" in d:
            print(f, "headline", d["value"], {k: (v.get("value"), (v.get("roofline") or {}).get("frac")) for k, v in d["workloads"].items()})
            s = d["workloads"]["slam_100k"].get("slam", {})
        print(f.split("/")[-1], {k: s.get(k) for k in ("pass_slam_s", "slam_s_median_timed", "icp_call_s", "localizer_host_s")})
PY
