#!/bin/bash
# Quick record of the headline workload: tools/ab_fast.sh OUTDIR TAG [extra bench args]
OUT=$1; TAG=$2; shift 2
mkdir -p $OUT
python bench.py --prepare-only
python bench.py --no-cpu-baseline --no-fixed30 "$@" 2>/dev/null | tail -1 > $OUT/bench_$TAG.json
python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json"))
r=d["roofline"]
print("$TAG:", round(d["value"],1), "scans/s", round(d["ms_per_step"],2), "ms/step; knn avg us", round(r["avg_launch_us"],1), "frac", round(r["frac"],4), "iters", round(d["mean_iterations"],2), {k:round(v["total_ms"]/r["profiled_steps"],2) for k,v in d["kernels"].items()})
PY
