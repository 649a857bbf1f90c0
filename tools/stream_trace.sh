#!/bin/bash
# kernel trace of the streaming workload: per-kernel totals and the matcher kernels of the LAST scan (a late-drive one)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-st}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $O/trace.log 2>&1
cd $R; python3 tools/trace_summary.py $O/trace | head -${2:-8}; rm -f $O/trace/*.db
python3 tools/timeline.py $O/trace | grep -E "knn|total"
tail -1 $O/trace.log | cut -c100-140
