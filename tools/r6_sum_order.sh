#!/bin/bash
# Round 6, item 5: what the scan-order walk of the reduction tree costs (headline, kernels per step), sorted order next to it
OUT=gpurun_out/r6s; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
for so in sorted scan sorted scan; do
  echo -n "== sum_order $so: "; rm -f bench_full.json
  python3 bench.py --sum-order $so --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/err.txt
  python3 -c "
import json; d=json.load(open('bench_full.json')); k=d['kernels']; r=d['roofline']
print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],2), 'ms; per step ms:', {n: round(v['total_ms']/r['profiled_steps'],3) for n,v in k.items() if n in ('p2plane_reduce','covariance','pretransform','knn_grid')})"
done 2>&1 | tee $OUT/sum_order.txt
