#!/bin/bash
TAG=${1:-fc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_slam.py tests/test_cpp_dropin.py tests/test_slam_replay.py -m gpu -x -q 2>&1 | tail -5
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for k in 1 2; do ./tools/slam_run $SEQ --filters sensor --mt > $OUT/mt_$k.json 2>> $OUT/err.log; python3 -c "import json; d=json.loads(open('$OUT/mt_$k.json').read().strip().splitlines()[-1]); print('MT 100k', d['scans_per_s'], 'wall', d['wall_s'], d['localizer_thread_s'], 'loops', d['loops_closed'], 'largest', d['largest_loop_batch'])"; done
tail -5 $OUT/err.log
