#!/bin/bash
# loop closing with smaller cells (PGICP_CELL_SCALE) and the ring counts / medium path that would take the longer queue; kernel totals per setting
OUT=gpurun_out/r6lc; mkdir -p $OUT
REPO=$(pwd)
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn pass us', r.get('avg_launch_us') and round(r['avg_launch_us'],1))"; }
{
for rep in 1 2; do
for s in "X=0" "PGICP_CELL_SCALE=0.7" "PGICP_CELL_SCALE=0.7 PGICP_FAST_RINGS_UNSEEDED=7 PGICP_FAST_RINGS_SEEDED=3" "PGICP_CELL_SCALE=0.7 PGICP_FAST_RINGS_UNSEEDED=9 PGICP_FAST_RINGS_SEEDED=4" "PGICP_CELL_SCALE=0.7 PGICP_MED_RINGS=8" "PGICP_CELL_SCALE=0.7 PGICP_FAST_RINGS_UNSEEDED=7 PGICP_FAST_RINGS_SEEDED=3 PGICP_MED_RINGS=8" "PGICP_CELL_SCALE=0.6 PGICP_FAST_RINGS_UNSEEDED=8 PGICP_FAST_RINGS_SEEDED=3" "PGICP_CELL_SCALE=0.8 PGICP_FAST_RINGS_UNSEEDED=6 PGICP_FAST_RINGS_SEEDED=3"; do
  echo -n "loop closing, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
done; done
} 2>&1 | tee $OUT/lc_cells.txt
cd /tmp && export TMPDIR=/tmp
for tag in base small; do
  if [ $tag = small ]; then export PGICP_CELL_SCALE=0.7; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_$tag -o t -- python3 $REPO/bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$OUT/trace_$tag.log 2>&1
  python3 $REPO/tools/trace_summary.py $REPO/$OUT/trace_$tag > $REPO/$OUT/trace_${tag}_summary.txt 2>&1
  rm -rf $REPO/$OUT/trace_$tag
done
cd $REPO
head -16 $OUT/trace_base_summary.txt; head -16 $OUT/trace_small_summary.txt
