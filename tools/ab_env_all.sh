#!/bin/bash
# A/B of environment settings on headline + stream + loop closing: tools/ab_env_all.sh "VAR=v" "VAR=w" ...
mkdir -p gpurun_out/r3
python3 bench.py --prepare-only > /dev/null 2>&1
for setting in "$@"; do
  tools/ab_headline.sh "$setting"
  echo -n "   stream: "; env $setting python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
  echo -n "   loop closing: "; env $setting python3 bench.py --workload loopclosure --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
done 2>&1 | tee gpurun_out/r3/ab_env_all.txt
