#!/bin/bash
# headline A/B over environment settings: tools/r4_ab_env.sh TAG "NAME=V[,NAME=V]" ...   ("base" = no setting)
OUT=gpurun_out/${1:-r4ab}; shift; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
for rep in 1 2; do
for spec in "$@"; do
  ( if [ "$spec" != base ]; then IFS=,; for kv in $spec; do export "$kv"; done; fi
    python3 bench.py --steps 8 --warmup 3 --no-workloads --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$spec', round(d['value'],1), round(d['ms_per_step'],3), 'fixed30', round(d['fixed_30_iterations']['scans_per_s'],1), 'sort', round(k['pretransform']['avg_us'],1), 'sel', round(k['trim_select']['avg_us'],1))" )
done; done | tee $OUT/summary.txt
