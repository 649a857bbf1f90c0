#!/bin/bash
# the loop-closure leg (512 pairs of 100 k-pt clouds, every pair its own 100 k-pt map with ~28 cm cells) under the matcher's knobs
OUT=gpurun_out/r6lk; mkdir -p $OUT
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn pass us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'set_map_ms', d.get('set_map_ms'))"; }
{
for rep in 1 2; do
for s in X=0 PGICP_CELL_SCALE=0.7 PGICP_CELL_SCALE=0.85 PGICP_CELL_SCALE=1.2 PGICP_FAST_RINGS_UNSEEDED=3 PGICP_FAST_RINGS_UNSEEDED=4 PGICP_FAST_RINGS_UNSEEDED=7 PGICP_FAST_RINGS_SEEDED=1 PGICP_FAST_RINGS_SEEDED=3 PGICP_MED_RINGS=2 PGICP_MED_RINGS=8 PGICP_KX=1 PGICP_KX=2 PGICP_KX=4 PGICP_SLOW_BLOCKS=4096; do
  echo -n "loop closing, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
done; done
} 2>&1 | tee $OUT/lc_knobs.txt
