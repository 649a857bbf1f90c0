#!/bin/bash
# the single-thread facade at sensor size under the library's tuning knobs (each launch of a scan is latency-bound there: the
# settings chosen for batches of 128 need not be the best for one problem)
OUT=gpurun_out/r6fk; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
one() { echo -n "$1: "; env $1 ./tools/slam_run $SEQ --filters sensor --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['localizer_host_s']; print(d['scans_per_s'], d['keyframes'], d['loops_closed'], d['map_rebuilds'], 'icp', h['icp'], 'probe', h['after_icp_parts']['overlap_probe'])"; }
{
for rep in 1 2; do
for s in X=0 PGICP_MED_RINGS=2 PGICP_MED_RINGS=4 PGICP_MED_RINGS=8 PGICP_MED_SHORT_RINGS=0 PGICP_MED_SHORT_RINGS=2 PGICP_FAST_RINGS_SEEDED=1 PGICP_FAST_RINGS_SEEDED=3 PGICP_FAST_RINGS_SEEDED=4 PGICP_FAST_RINGS_UNSEEDED=3 PGICP_FAST_RINGS_UNSEEDED=7 PGICP_GRAPH_MAX_P=1 PGICP_SLOW_BLOCKS=512 PGICP_SLOW_BLOCKS=4096 PGICP_FAST_LANES=8 PGICP_FAST_LANES=32 PGICP_POLL_US=2000 PGICP_SEL_BAND=0 PGICP_SLOW_SQUARE_ROWS=0; do one $s; done
done
} 2>&1 | tee $OUT/knobs.txt
