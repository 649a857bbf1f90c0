#!/bin/bash
# after a kernel-side change: the GPU suite, then the default line (the driver's command)
TAG=${1:-ck}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; grep -E "passed|failed|error" $OUT/gputest.log | tail -3
python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
python3 - $OUT/bench_n1.json <<'PY'
import json, sys
ln = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(ln)
print("line bytes", len(ln), "tail:", ln[-700:])
print("headline", d["value"], d["roofline"]["frac"], "wall", d["command_wall_s"])
for k, v in d["workloads"].items():
    print(" ", k, v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("error"), round(v.get("leg_wall_s", 0), 1))
PY
