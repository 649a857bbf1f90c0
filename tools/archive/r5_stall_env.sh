#!/bin/bash
# which part of an ICP call holds the 25-40 ms stalls of a young process (PGICP_HOST_TIMING), and do they need the interrupt path?
TAG=${1:-se}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
run() { # label, env...
  local label=$1; shift
  for k in 1 2 3; do
    sleep 3
    env "$@" PGICP_HOST_TIMING=1 ./tools/slam_run $SEQ --filters sensor --passes 1 --limit 200 > $OUT/${label}_$k.json 2> $OUT/${label}_$k.err
    python - $OUT/${label}_$k.json $OUT/${label}_$k.err $label$k <<'PY'
import json, sys, re
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
slow = []
for ln in open(sys.argv[2]):
    m = re.match(r"align_batch P=(\d+): begin ([\d.]+) ms, iterations ([\d.]+) ms, tail ([\d.]+) ms", ln)
    if m and max(float(m.group(2)), float(m.group(3)), float(m.group(4))) > 5:
        slow.append(tuple(float(m.group(i)) for i in (2, 3, 4)))
print(sys.argv[3], "icp_s", d["localizer_host_s"]["icp"], "filters_s", d["localizer_host_s"]["filters_and_sensor_transform"], "after_s", d["localizer_host_s"]["after_icp"], "stalled calls (begin, iterations, tail ms):", slow[:12], len(slow))
PY
  done
}
run plain A=1
run nointr HSA_ENABLE_INTERRUPT=0
run poll_long PGICP_POLL_US=100000
