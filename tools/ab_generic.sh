#!/bin/bash
# A/B of one environment knob on stream (1 vehicle, fleet of 16), loop closing and the headline: tools/ab_generic.sh VAR "v1 v2 ..."
VAR=$1; VALS=$2
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', sys.argv[1], round(d['value'],1))" "$1"; }
for v in $VALS; do
  export $VAR=$v; echo "== $VAR=$v"
  python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | val stream
  python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | val stream
  python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | val fleet16
  python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | val loopclosure
  python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-profile 2>/dev/null | tail -1 | val headline
done
