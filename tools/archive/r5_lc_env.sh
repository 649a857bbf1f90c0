#!/bin/bash
# environment knobs on the loop-closure leg: tools/r5_lc_env.sh "VAR=v ..." ...
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
mkdir -p gpurun_out/lc_env
for setting in "X=1" "$@" "X=1"; do
  echo -n "== $setting: "
  env $setting python3 bench.py --workload loopclosure --pairs 512 --steps 3 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
done 2>&1 | tee gpurun_out/lc_env/result.txt
