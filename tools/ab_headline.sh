#!/bin/bash
# A/B of environment knobs on the headline only: tools/ab_headline.sh "VAR=v VAR2=w" "VAR=x" ...  (one setting per argument)
for setting in "$@"; do
  echo -n "== $setting: "
  env $setting python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads 2>/tmp/ab_err.txt | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; r=d['roofline']
print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],2), 'ms; per step ms:', {n: round(v['total_ms']/r['profiled_steps'],2) for n,v in k.items()})"
done
