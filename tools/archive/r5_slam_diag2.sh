#!/bin/bash
# round 5: cold single passes of slam_run as separate processes (allocation calls timed, slowest ICP calls listed), the tie census
TAG=${1:-sd2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for k in 1 2 3; do ./tools/slam_run $SEQ --filters sensor --passes 1 > $OUT/cold_$k.json 2>> $OUT/err.log; sleep 2; done
./tools/slam_run $SEQ --filters sensor --passes 4 > $OUT/warm.json 2>> $OUT/err.log
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], {k: d.get(k) for k in ("pass_slam_s", "localizer_host_s", "alloc_calls_and_seconds_pass0", "alloc_calls_and_seconds_last_pass", "slowest_icp_calls_pass0", "icp_call_s")})
PY
python -m pytest tests/test_sensitivity.py -m gpu -x -q -s 2>&1 | tail -5
python tools/tie_census.py --out $OUT/tie_census.json 2>&1 | tail -3
