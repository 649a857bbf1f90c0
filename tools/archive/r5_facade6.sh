#!/bin/bash
TAG=${1:-ff}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_slam.py tests/test_gpu_filters.py tests/test_cpp_dropin.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for e in 0 1; do
  if [ $e = 1 ]; then export PGSLAM_SYNC_HOST_COMPACTION=1; else unset PGSLAM_SYNC_HOST_COMPACTION; fi
  ./tools/slam_run $SEQ --filters sensor --passes 4 > $OUT/st_$e.json 2>> $OUT/err.log
  python3 - $OUT/st_$e.json $e <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('sync compaction =', sys.argv[2], d['pass_slam_s'], 599 / d['slam_s_median_timed'], 'scans/s', d['localizer_host_s']['filters_and_sensor_transform'], d['localizer_host_s']['icp'], d['localizer_host_s']['after_icp'], d['tracking_error_rms_m'], d['keyframes'])
PY
done
unset PGSLAM_SYNC_HOST_COMPACTION
for k in 1 2; do ./tools/slam_run $SEQ --filters sensor --mt > $OUT/mt_$k.json 2>> $OUT/err.log; python3 -c "import json; d=json.loads(open('$OUT/mt_$k.json').read().strip().splitlines()[-1]); print('MT 100k', d['scans_per_s'], 'stage thread', d['input_stage_thread_s'], d['localizer_thread_s'])"; done
tail -3 $OUT/err.log
