#!/bin/bash
# full timeline of the last streamed scan (every kernel: start, duration, gap) + per-kernel totals
R=$PWD; O=$R/gpurun_out/${1:-r4st}; mkdir -p $O
python3 $R/bench.py --workload stream --prepare-only > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-input > $O/trace.log 2>&1
cd $R; python3 tools/trace_summary.py $O/trace | head -40 > $O/summary.txt; rm -f $O/trace/*.db
python3 tools/timeline.py $O/trace > $O/timeline_last_scan.txt
tail -1 $O/trace.log | cut -c1-200
rm -rf $O/trace
head -40 $O/summary.txt; tail -75 $O/timeline_last_scan.txt
