#!/bin/bash
# look-ahead of a single problem's iterations: PGICP_SPECULATE unset (by size: whole iteration up to 32 k points, matcher pass above), 0 (off), 1 (matcher pass), 2 (whole iteration)
OUT=gpurun_out/r6sp2; mkdir -p $OUT
{
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -4
PGICP_SPECULATE=2 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bit_exact.py tests/test_gpu_edge_cases.py tests/test_gpu_matcher_state.py tests/test_local_mapper.py tests/test_slam_replay.py tests/test_slam.py -m gpu -x -q 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -3
python3 bench.py --workload slam --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
for rep in 1 2 3; do for s in X=0 PGICP_SPECULATE=0 PGICP_SPECULATE=1 PGICP_SPECULATE=2; do
  echo -n "facade 10k, $s: "; env $s ./tools/slam_run /tmp/pgslam_amd_seq_4500_10000_0.8.bin --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['localizer_host_s']; print(d['scans_per_s'], d['keyframes'], d['loops_closed'], 'icp', h['icp'], 'probe', h['after_icp_parts']['overlap_probe'])"
  echo -n "facade 10k mt, $s: "; env $s ./tools/slam_run /tmp/pgslam_amd_seq_4500_10000_0.8.bin --mt --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scans_per_s'], d['keyframes'], d['loops_closed'])"
  echo -n "facade 100k, $s: "; env $s ./tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['localizer_host_s']; print(d['scans_per_s'], d['keyframes'], d['loops_closed'], 'icp', h['icp'])"
done; done
} 2>&1 | tee $OUT/spec2.txt
