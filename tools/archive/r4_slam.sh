#!/bin/bash
# SLAM facade on the GPU box: its tests, then the 100 k-pt leg with the local maps in device memory / through the host, and the 10 k-pt leg
TAG=${1:-s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_filters.py tests/test_slam.py tests/test_cpp_dropin.py tests/test_slam_replay.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --slam-record 8 --steps 1 --warmup 0 > $OUT/bench_slam100k.json 2> $OUT/bench.err
PGSLAM_HOST_LOCAL_MAP=1 python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --slam-record 8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_slam100k_hostmap.json 2>> $OUT/bench.err
python bench.py --workload slam --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_slam10k.json 2>> $OUT/bench.err
PGSLAM_HOST_LOCAL_MAP=1 python bench.py --workload slam --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_slam10k_hostmap.json 2>> $OUT/bench.err
tail -3 $OUT/gputest.log
for f in $OUT/bench_slam*.json; do python3 - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d["slam"]
    print(sys.argv[1].split("/")[-1], "%.1f scans/s" % d["value"], s["localizer_host_s"], "rebuilds", s["map_rebuilds"], "on device", s.get("device_map_rebuilds"), "kf", s["keyframes"], "err", s["tracking_error_rms_m"])
except Exception as e:
    print(sys.argv[1], "failed", type(e).__name__, e)
PY
done
