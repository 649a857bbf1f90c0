#!/bin/bash
# Round 6, item 3: the succinct cell table (MapDev::sw) against the dense tables, per leg.  PGICP_TABLES=dense|succinct|auto
OUT=gpurun_out/r6t; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn launch us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'frac', r.get('frac') and round(r['frac'],4), 'set_map_ms', d.get('set_map_ms'))"; }
{
for t in dense succinct dense succinct; do
  echo -n "loop closing, PGICP_TABLES=$t: "; rm -f bench_full.json
  PGICP_TABLES=$t python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
done
for t in dense succinct; do
  echo -n "headline, PGICP_TABLES=$t: "; rm -f bench_full.json
  PGICP_TABLES=$t python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/err.txt; val
done
for t in dense succinct dense succinct; do
  echo -n "stream (one vehicle, 2 M-pt sliding map), PGICP_TABLES=$t: "; rm -f bench_full.json
  PGICP_TABLES=$t python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
done
} 2>&1 | tee $OUT/tables.txt
