#!/bin/bash
# A/B of one environment knob on the streaming workload (kernel-trace totals of the matcher kernels and the bench value):
# tools/ab_stream.sh VAR "v1 v2 ..."
VAR=$1; VALS=$2
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in $VALS; do
  O=$R/gpurun_out/ab_$v; mkdir -p $O; export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/trace.log 2>&1
  echo "== $VAR=$v: $(tail -1 $O/trace.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "scans/s (under the profiler)")')"
  (cd $R; python3 tools/trace_summary.py $O/trace | grep -E "k_knn|k_sel" ; rm -f $O/trace/*.db)
  python3 $R/bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("   plain run:", round(d["value"],1), "scans/s")'
done
