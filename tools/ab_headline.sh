#!/bin/bash
# A/B of environment knobs on the headline only: tools/ab_headline.sh "VAR=v VAR2=w" "VAR=x" ...  (one setting per argument)
# (the figures come from the full record bench.py writes -- bench_full.json --, the printed line is the short form)
for setting in "$@"; do
  echo -n "== $setting: "
  rm -f bench_full.json
  env $setting python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/ab_err.txt
  python3 -c "
import json; d=json.load(open('bench_full.json')); k=d['kernels']; r=d['roofline']
print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],2), 'ms; frac', round(r['frac'],4), 'knn launch us', round(r['avg_launch_us'],1), '; per step ms:', {n: round(v['total_ms']/r['profiled_steps'],2) for n,v in k.items()})"
done
