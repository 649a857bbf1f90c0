#!/bin/bash
# round 3: the whole GPU suite, then the default bench line (headline + workloads legs)
mkdir -p gpurun_out/r3
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -15 | tee gpurun_out/r3/tests_all.txt
( time python3 bench.py --steps 20 --warmup 5 ) > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
tail -c 3000 gpurun_out/r3/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "wall", d["command_wall_s"])
for k,v in (d.get("workloads") or {}).items():
    print(k, {a: v.get(a) for a in ("value","unit","leg_wall_s","error","host_input")}, "roofline", (v.get("roofline") or {}).get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
print("loop_closure", {a: d["loop_closure"].get(a) for a in ("value","rccl_ranks_seen","speedup_vs_one_gpu")} if d.get("loop_closure") else None)
PY
