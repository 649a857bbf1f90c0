#!/bin/bash
# Round 6, item 4: the façade's overlap probe seeded from the ICP's correspondences (PGSLAM_PROBE_SEEDS=1, the default) against
# the unseeded probe (=0): parity tests first, then the sensor-size drive both ways (single-thread and three-thread façade), then
# one scan's launch timeline.
OUT=gpurun_out/r6p; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
REPO=$(pwd)
{
python3 -m pytest tests/test_gpu_parity.py tests/test_slam_replay.py -m gpu -x -q 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -8
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
for rep in 1 2; do for s in 0 1; do
  echo -n "single-thread, PGSLAM_PROBE_SEEDS=$s: "; PGSLAM_PROBE_SEEDS=$s ./tools/slam_run $SEQ --filters sensor --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('scans_per_s','keyframes','loops_closed','map_rebuilds','overlap_probes_seeded','tracking_error_rms_m','keyframe_error_rms_m','tracking_error_last_m')}, d['localizer_host_s']['after_icp_parts'])"
  echo -n "three threads, PGSLAM_PROBE_SEEDS=$s: "; PGSLAM_PROBE_SEEDS=$s ./tools/slam_run $SEQ --filters sensor --mt --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('scans_per_s','keyframes','loops_closed','map_rebuilds','overlap_probes_seeded','tracking_error_rms_m')})"
done; done
} 2>&1 | tee $OUT/probe_seeds.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace -o t -- $REPO/tools/slam_run $SEQ --filters sensor --limit 120 > $REPO/$OUT/trace.log 2>&1
cd $REPO
python3 tools/scan_timeline.py $OUT/trace 5 > $OUT/scan_timeline.txt 2>&1
rm -rf $OUT/trace
tail -100 $OUT/scan_timeline.txt
