#!/bin/bash
# the headline under cell size x ring counts x medium path (round 4's sweep, on round 6's kernels)
OUT=gpurun_out/r6hc; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn pass us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'unseeded', r.get('avg_unseeded_launch_us') and round(r['avg_unseeded_launch_us'],1), 'seeded', r.get('avg_seeded_launch_us') and round(r['avg_seeded_launch_us'],1), 'set_map_ms', d.get('set_map_ms'))"; }
{
for rep in 1 2; do
for s in "X=0" "PGICP_CELL_SCALE=0.85" "PGICP_CELL_SCALE=0.7" "PGICP_CELL_SCALE=0.7 PGICP_FAST_RINGS_UNSEEDED=5 PGICP_FAST_RINGS_SEEDED=2" "PGICP_CELL_SCALE=0.7 PGICP_FAST_RINGS_UNSEEDED=4 PGICP_FAST_RINGS_SEEDED=2 PGICP_MED_RINGS=8" "PGICP_CELL_SCALE=0.85 PGICP_FAST_RINGS_UNSEEDED=4 PGICP_FAST_RINGS_SEEDED=2" "PGICP_CELL_SCALE=0.85 PGICP_MED_RINGS=8" "PGICP_MED_RINGS=8" "PGICP_FAST_RINGS_SEEDED=2" "PGICP_CELL_SCALE=0.6 PGICP_FAST_RINGS_UNSEEDED=6 PGICP_FAST_RINGS_SEEDED=2 PGICP_MED_RINGS=8" "PGICP_CELL_SCALE=1.15"; do
  echo -n "headline, $s: "; rm -f bench_full.json; env $s python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/err.txt; val
done; done
} 2>&1 | tee $OUT/headline_cells.txt
