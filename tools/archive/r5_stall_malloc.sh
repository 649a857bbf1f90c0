#!/bin/bash
# hypothesis: the pauses are KFD evicting the process's queues when glibc munmap()s (or trims) host memory that HIP had pinned
# on the fly for a pageable hipMemcpyAsync (MMU-notifier invalidation of a userptr BO); glibc's dynamic mmap threshold stops
# using mmap for that size after the first few frees -- hence "young process only"
TAG=${1:-sm}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
run() {
  local label=$1; shift
  for k in 1 2 3; do
    sleep 3
    env "$@" PGICP_HOST_TIMING=1 ./tools/slam_run $SEQ --filters sensor --passes 1 --limit 200 > $OUT/${label}_$k.json 2> $OUT/${label}_$k.err
    echo "$label $k: stalled align calls: $(grep align_batch $OUT/${label}_$k.err | awk '{ if ($6+0 > 5) c++ } END { print c+0 }')  $(python3 -c "import json,sys; d=json.loads(open('$OUT/${label}_$k.json').read().strip().splitlines()[-1]); print(d['localizer_host_s'])")"
  done
}
run plain A=1
run no_mmap MALLOC_MMAP_THRESHOLD_=33554432 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=268435456
run plain_again A=1
