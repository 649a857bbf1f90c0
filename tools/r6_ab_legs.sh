#!/bin/bash
# A/B of one environment setting over the bench legs (interleaved, each leg twice): usage r6_ab_legs.sh "ENV=VAL"
OUT=gpurun_out/r6ab; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn pass us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'frac', r.get('frac') and round(r['frac'],4))"; }
slam() { echo -n "facade 100k single thread, $1: "; env $1 ./tools/slam_run $SEQ --filters sensor --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['localizer_host_s']; print(d['scans_per_s'], d['keyframes'], d['loops_closed'], d['map_rebuilds'], 'icp', h['icp'], 'probe', h['after_icp_parts']['overlap_probe'])"; }
{
for rep in 1 2; do for s in X=0 $1; do
  echo -n "headline, $s: "; rm -f bench_full.json; env $s python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/err.txt; val
  echo -n "loop closing, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
  echo -n "stream 1, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "stream fleet 16, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  slam $s
done; done
} 2>&1 | tee $OUT/ab_legs.txt
