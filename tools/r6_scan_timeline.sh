#!/bin/bash
# one scan of the facade at sensor size, launch by launch (where the launches of a scan go)
OUT=gpurun_out/r6st; mkdir -p $OUT
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -o t -- $R/tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --limit 120 > $R/$OUT/trace.log 2>&1
cd $R
python3 tools/scan_timeline.py $OUT/trace 5 > $OUT/scan_timeline.txt 2>&1
rm -rf $OUT/trace
tail -120 $OUT/scan_timeline.txt
