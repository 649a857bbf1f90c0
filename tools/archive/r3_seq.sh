#!/bin/bash
# per-launch durations of the matcher kernels of the headline's last step + per-kernel totals (rocprofv3 kernel trace)
R=$PWD; O=$R/gpurun_out/r3/seq; mkdir -p $O
python3 bench.py --prepare-only > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-fixed30 --no-cpu-baseline --no-host-input --no-workloads > $O/trace.log 2>&1
cd $R
python3 tools/knn_seq.py $O/trace 36 | tee $O/knn_seq.txt
python3 tools/trace_summary.py $O/trace | head -12 | tee $O/summary.txt
rm -rf $O/trace
