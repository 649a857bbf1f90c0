#!/bin/bash
# interleaved A/B of one setting of the single-thread facade at sensor size: usage r6_facade_ab.sh "ENV=VAL" [reps]
OUT=gpurun_out/r6fa; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
one() { echo -n "$1: "; env $1 ./tools/slam_run $SEQ --filters sensor --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['localizer_host_s']; print(d['scans_per_s'], d['keyframes'], d['loops_closed'], d['map_rebuilds'], 'icp', h['icp'], 'probe', h['after_icp_parts']['overlap_probe'])"; }
{
for rep in $(seq 1 ${2:-6}); do one X=0; for s in $1; do one $s; done; done
} 2>&1 | tee $OUT/ab.txt
