#!/bin/bash
# kernel totals of the headline at several grid cell scales / ring settings: "name:ENV=..,ENV=.." per argument after the tag
OUT=gpurun_out/${1:-r4cst}; shift; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do export "$kv"; done
    python3 $R/bench.py --steps 4 --warmup 2 --no-workloads --no-cpu-baseline --no-host-input --no-fixed30 2>/dev/null | tail -1 > $R/$OUT/b_$name.json
    rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$name -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-workloads --no-cpu-baseline --no-host-input --no-fixed30 --no-profile > /dev/null 2>&1 )
  python3 $R/tools/db_summary.py $R/$OUT/prof_$name/trace_results.db 16 > $R/$OUT/trace_$name.txt 2>&1
  rm -rf $R/$OUT/prof_$name
  python3 - $R/$OUT/b_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
    print("%s: %.0f scans/s  %.2f ms/step  unseeded %.0f us  seeded %.0f us" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_unseeded_launch_us", 0), r.get("avg_seeded_launch_us", 0)))
except Exception as e:
    print(sys.argv[2], "failed", type(e).__name__)
PY
  head -12 $R/$OUT/trace_$name.txt
done 2>&1 | tee $R/$OUT/summary.txt
