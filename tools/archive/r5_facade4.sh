#!/bin/bash
# how much of the input stage is the host compaction?  sensor filters (a handful of points dropped per scan) against identity filters (none)
TAG=${1:-fd}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for f in sensor identity sensor identity; do
  ./tools/slam_run $SEQ --filters $f --passes 3 > $OUT/st_$f.json 2>> $OUT/err.log
  python3 - $OUT/st_$f.json $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['pass_slam_s'], d['localizer_host_s']['filters_and_sensor_transform'], d['localizer_host_s']['icp'], d['localizer_host_s']['after_icp'], 'points after filters', d['points_after_filters_last_scan'])
PY
done
