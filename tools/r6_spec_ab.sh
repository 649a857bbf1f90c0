#!/bin/bash
# the next iteration's matcher pass enqueued ahead of the convergence flag (one problem per launch; PGICP_SPECULATE=0: off): parity, then the legs
OUT=gpurun_out/r6sp; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
{
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -4
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step')"; }
for rep in 1 2 3; do for s in X=0 PGICP_SPECULATE=0; do
  echo -n "stream 1, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "stream 4, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload stream --streams 4 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "facade 100k, $s: "; env $s ./tools/slam_run $SEQ --filters sensor --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['localizer_host_s']; print(d['scans_per_s'], d['keyframes'], d['loops_closed'], 'icp', h['icp'])"
  echo -n "facade 100k mt, $s: "; env $s ./tools/slam_run $SEQ --filters sensor --mt --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scans_per_s'], d['keyframes'], d['loops_closed'])"
  echo -n "facade 10k, $s: "; env $s ./tools/slam_run /tmp/pgslam_amd_seq_4500_10000_0.8.bin --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scans_per_s'], d['keyframes'], d['loops_closed'])"
done; done
} 2>&1 | tee $OUT/spec.txt
