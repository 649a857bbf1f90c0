#!/bin/bash
# loop closing (512 pairs of 100k-pt clouds, every pair its own 100k-pt map): the grid's cell size, the x refinement, the rings of the fast and medium paths
OUT=gpurun_out/${1:-lck}; mkdir -p $OUT
run() { env "$@" python3 bench.py --workload loopclosure --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step; fast matcher avg', round(r['avg_launch_us']), 'us, ok', d['pairs_ok'], 'acc', d['pairs_accepted'], 'iters', round(d['mean_iterations'],3))"; }
{
run A=1
for s in 0.6 0.71 0.85; do
  run PGICP_CELL_SCALE=$s PGICP_FAST_RINGS_UNSEEDED=4
  run PGICP_CELL_SCALE=$s PGICP_FAST_RINGS_UNSEEDED=5 PGICP_FAST_RINGS_SEEDED=2
  run PGICP_CELL_SCALE=$s PGICP_FAST_RINGS_UNSEEDED=5 PGICP_FAST_RINGS_SEEDED=2 PGICP_MED_RINGS=6
  run PGICP_CELL_SCALE=$s PGICP_FAST_RINGS_UNSEEDED=6 PGICP_FAST_RINGS_SEEDED=2 PGICP_MED_RINGS=8
done
run PGICP_FAST_RINGS_UNSEEDED=5 PGICP_FAST_RINGS_SEEDED=2
run A=2
} 2>&1 | tee $OUT/summary.txt
