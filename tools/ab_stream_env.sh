#!/bin/bash
# stream value under environment settings, three runs each: tools/ab_stream_env.sh "VAR=v" ...
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
for setting in "$@"; do
  echo -n "$setting stream: "
  for i in 1 2 3; do env $setting python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1), end=' ')"; done; echo
done 2>&1 | tee gpurun_out/r3/ab_stream_env.txt
