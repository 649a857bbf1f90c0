#!/bin/bash
# A/B of one environment knob on the streaming and loop-closing workloads: tools/ab_stream.sh VAR "v1 v2 ..."
VAR=$1; VALS=$2
mkdir -p gpurun_out/ab
val() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d['value'],1), d.get('mean_iterations'))" $1 $2; }
for v in $VALS; do
  for w in 1 2; do env $VAR=$v python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab/s.json; val gpurun_out/ab/s.json "$VAR=$v stream"; done
  env $VAR=$v python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > gpurun_out/ab/l.json; val gpurun_out/ab/l.json "$VAR=$v loopclosure"
done
