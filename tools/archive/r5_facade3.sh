#!/bin/bash
TAG=${1:-fc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for e in 0 1; do
  ./tools/slam_run $SEQ --filters sensor --mt > $OUT/mt_$e.json 2>> $OUT/err.log
  python3 - $OUT/mt_$e.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('MT 100k', d['scans_per_s'], 'wall', d['wall_s'], 'input-stage thread busy', d['input_stage_thread_s'], d['localizer_thread_s'], 'loops', d['loops_closed'])
PY
done
tail -3 $OUT/err.log
