#!/bin/bash
# the wave-per-query path's prune radius (PGICP_PRUNE_PCT: the "no neighbour" proof covers maxDist x (1 + pct/100)) on the legs where scan points
# run ahead of the map: streaming (one vehicle, fleet of 16) and the facade at sensor size
OUT=gpurun_out/r6pp; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step')"; }
{
for rep in 1 2; do for p in 3 5 8 12; do
  echo -n "stream 1, PGICP_PRUNE_PCT=$p: "; rm -f bench_full.json; PGICP_PRUNE_PCT=$p python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "stream fleet 16, PGICP_PRUNE_PCT=$p: "; rm -f bench_full.json; PGICP_PRUNE_PCT=$p python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "facade 100k, PGICP_PRUNE_PCT=$p: "; PGICP_PRUNE_PCT=$p ./tools/slam_run $SEQ --filters sensor --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scans_per_s'], d['keyframes'], d['loops_closed'])"
done; done
} 2>&1 | tee $OUT/prune.txt
