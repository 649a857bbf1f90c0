#!/bin/bash
TAG=${1:-fa}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
./tools/slam_run $SEQ --filters sensor --passes 4 > $OUT/warm.json 2>> $OUT/err.log
python3 -c "import json; d=json.loads(open('$OUT/warm.json').read().strip().splitlines()[-1]); print('warm', d['pass_slam_s'], json.dumps(d['localizer_host_s']), 'kf', d['keyframes'], 'rebuilds', d['map_rebuilds'], 'loops', d['loops_closed'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $OLDPWD/tools/slam_run $SEQ --filters sensor --passes 2 > $OUT/traced.json 2>> $OUT/err.log
cd $OLDPWD
python tools/trace_summary.py $OUT/trace > $OUT/trace_summary.txt 2>&1; head -45 $OUT/trace_summary.txt
rm -rf $OUT/trace
