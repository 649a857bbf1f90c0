#!/bin/bash
# a cold slam_run pass under rocprofv3 (kernel + HIP runtime trace): what are the 25-40 ms stalls?
TAG=${1:-st}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
R=$PWD
./tools/slam_run $SEQ --filters sensor --passes 1 --limit 120 > $OUT/plain_1.json 2>> $OUT/err.log
sleep 3
./tools/slam_run $SEQ --filters sensor --passes 1 --limit 120 > $OUT/plain_2.json 2>> $OUT/err.log
sleep 3
cd /tmp && export TMPDIR=/tmp
for k in 1 2; do
  rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/trace_$k -o t -- $R/tools/slam_run $SEQ --filters sensor --passes 1 --limit 120 > $OUT/traced_$k.json 2>> $OUT/err.log
  sleep 3
done
cd $R
for k in 1 2; do python tools/stall_report.py $OUT/trace_$k > $OUT/stall_report_$k.txt 2>&1; done
python - $OUT <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], {k: d.get(k) for k in ("pass_slam_s", "localizer_host_s", "icp_call_s")}, [x["scan"] for x in d.get("slowest_icp_calls_pass0", []) if x["s"] > 0.005])
PY
head -70 $OUT/stall_report_1.txt; head -40 $OUT/stall_report_2.txt
rm -rf $OUT/trace_1 $OUT/trace_2
python -m pytest tests/test_gpu_filters.py tests/test_cpp_dropin.py -m gpu -x -q 2>&1 | tail -5
