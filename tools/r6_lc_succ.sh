#!/bin/bash
# loop closing: smaller cells WITH the succinct cell table (the dense tables' build grows with the cell count: k_mfill, k_near_b)
OUT=gpurun_out/r6ls; mkdir -p $OUT
REPO=$(pwd)
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn pass us', r.get('avg_launch_us') and round(r['avg_launch_us'],1))"; }
R="PGICP_FAST_RINGS_UNSEEDED=7 PGICP_FAST_RINGS_SEEDED=3 PGICP_MED_RINGS=8"
{
for rep in 1 2; do
for s in "X=0" "PGICP_TABLES=succinct" "PGICP_TABLES=succinct PGICP_CELL_SCALE=0.7" "PGICP_TABLES=succinct PGICP_CELL_SCALE=0.7 $R" "PGICP_TABLES=succinct PGICP_CELL_SCALE=0.6 $R" "PGICP_TABLES=succinct PGICP_CELL_SCALE=0.8 $R" "PGICP_TABLES=succinct PGICP_CELL_SCALE=0.7 PGICP_FAST_RINGS_UNSEEDED=9 PGICP_FAST_RINGS_SEEDED=4 PGICP_MED_RINGS=8" "PGICP_TABLES=succinct PGICP_CELL_SCALE=0.5 PGICP_FAST_RINGS_UNSEEDED=10 PGICP_FAST_RINGS_SEEDED=4 PGICP_MED_RINGS=8" "PGICP_CELL_SCALE=0.7 $R"; do
  echo -n "loop closing, $s: "; rm -f bench_full.json; env $s python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
done; done
} 2>&1 | tee $OUT/lc_succ.txt
cd /tmp && export TMPDIR=/tmp
export PGICP_TABLES=succinct PGICP_CELL_SCALE=0.7 $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -o t -- python3 $REPO/bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$OUT/trace.log 2>&1
python3 $REPO/tools/trace_summary.py $REPO/$OUT/trace > $REPO/$OUT/trace_summary.txt 2>&1
rm -rf $REPO/$OUT/trace
cd $REPO
head -24 $OUT/trace_summary.txt
