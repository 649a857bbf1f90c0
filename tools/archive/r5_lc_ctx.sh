#!/bin/bash
OUT=gpurun_out/${1:-lcx}; mkdir -p $OUT
for c in 1 2 3 4 1 2; do python bench.py --workload loopclosure --steps 4 --warmup 2 --no-cpu-baseline --no-profile --lc-contexts $c 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('contexts $c', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms', d['pairs_ok'], d['pairs_accepted'])"; done
