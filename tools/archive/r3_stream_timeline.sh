#!/bin/bash
# full timeline of the last streamed scan (every kernel: start, duration, gap) + per-kernel totals
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3/stream_trace; mkdir -p $O
python3 $R/bench.py --workload stream --prepare-only > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-input > $O/trace.log 2>&1
cd $R; python3 tools/trace_summary.py $O/trace | head -30 | tee $O/summary.txt; rm -f $O/trace/*.db
python3 tools/timeline.py $O/trace | tee $O/timeline_last_scan.txt | tail -70
tail -1 $O/trace.log | cut -c1-200
rm -rf $O/trace
