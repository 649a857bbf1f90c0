#!/bin/bash
# Round 6, item 4: rings the capped seeded probe's fast pass walks (PGICP_PROBE_RINGS) -- tests, then the sensor-size drive per setting
OUT=gpurun_out/r6pr; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
{
python3 -m pytest tests/test_gpu_parity.py tests/test_slam_replay.py -m gpu -x -q 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -8
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
for rep in 1 2; do for r in 0 1 2 3; do
  echo -n "single-thread, PGICP_PROBE_RINGS=$r: "; PGICP_PROBE_RINGS=$r ./tools/slam_run $SEQ --filters sensor --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('scans_per_s','keyframes','loops_closed','map_rebuilds','overlap_probes_seeded','tracking_error_rms_m')}, d['localizer_host_s']['after_icp_parts']['overlap_probe'])"
done; done
} 2>&1 | tee $OUT/probe_rings.txt
