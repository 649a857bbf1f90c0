#!/bin/bash
# kernel trace of the C++ SLAM driver over the first scans of the configs[3] sequence: GPU busy share, kernel totals, and the
# timeline of the last scan's kernels
R=$PWD
python3 bench.py --workload slam --prepare-only --slam-scans ${1:-1500} > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" 2>/dev/null
SEQ=/tmp/pgslam_amd_seq_${1:-1500}_10000_0.8.bin
O=$R/gpurun_out/${2:-r4slamtrace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/slam_run $SEQ | tail -1 | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $R/tools/slam_run $SEQ > $O/trace.log 2>&1
cd $R; python3 tools/trace_summary.py $O/trace | head -36 > $O/summary.txt; rm -f $O/trace/*.db
python3 tools/timeline.py $O/trace > $O/timeline_last_scan.txt 2>&1
python3 - <<PY >> $O/summary.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0])))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)
print("kernels:", len(rows), "busy %.1f ms of %.1f ms span (%.0f %%)" % (busy / 1e6, span / 1e6, 100.0 * busy / span))
PY
rm -rf $O/trace
cat $O/summary.txt | cut -c1-120; tail -60 $O/timeline_last_scan.txt
