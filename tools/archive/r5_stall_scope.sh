#!/bin/bash
TAG=${1:-ss}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for k in 1 2 3 4; do
  sleep 3
  PGICP_PROFILE_ALL=1 PGICP_STALL_LOG=5 PGICP_HOST_TIMING=1 ./tools/slam_run $SEQ --filters sensor --passes 1 --limit 200 > $OUT/run_$k.json 2> $OUT/run_$k.err
  echo "run $k: $(grep -c 'pgicp stall' $OUT/run_$k.err) stall lines; slow align calls: $(grep align_batch $OUT/run_$k.err | awk '{ if ($6+0 > 5) c++ } END { print c+0 }')"
  grep 'pgicp stall' $OUT/run_$k.err | sed 's/(context.*//' | sort | uniq -c | sort -rn | head -12
done
rocm-smi --showclocks 2>/dev/null | head -20
