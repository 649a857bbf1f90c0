#!/bin/bash
# A/B of the fast matcher kernels on the headline workload: tools/ab_fast.sh OUTDIR "0 1 2" [extra bench args]
OUT=$1; KS=$2; shift 2
mkdir -p $OUT
python bench.py --prepare-only
for k in $KS; do
  PGICP_FAST_KERNEL=$k python bench.py --no-cpu-baseline --no-fixed30 "$@" 2>/dev/null | tail -1 > $OUT/bench_k$k.json
  python - <<PY
import json
d=json.load(open("$OUT/bench_k$k.json"))
r=d["roofline"]
print("fast_kernel $k:", round(d["value"],1), "scans/s", round(d["ms_per_step"],2), "ms/step; knn avg us", round(r["avg_launch_us"],1), "frac", round(r["frac"],4), "iters", round(d["mean_iterations"],2), {k:round(v["total_ms"]/r["profiled_steps"],2) for k,v in d["kernels"].items()})
PY
done
