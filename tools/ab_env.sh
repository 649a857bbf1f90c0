#!/bin/bash
# A/B of one environment knob on the headline workload: tools/ab_env.sh OUTDIR VAR "v1 v2 ..." [extra bench args]
OUT=$1; VAR=$2; VALS=$3; shift 3
mkdir -p $OUT
python bench.py --prepare-only
for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-fixed30 --no-host-input "$@" 2>/dev/null | tail -1 > $OUT/bench_${VAR}_$v.json
  python - <<PY
import json
d=json.load(open("$OUT/bench_${VAR}_$v.json"))
r=d["roofline"]
print("$VAR=$v:", round(d["value"],1), "scans/s", round(d["ms_per_step"],2), "ms/step; knn avg us", round(r["avg_launch_us"],1), "frac", round(r["frac"],4), "iters", round(d["mean_iterations"],2), {k:round(v["total_ms"]/r["profiled_steps"],2) for k,v in d["kernels"].items()})
PY
done
