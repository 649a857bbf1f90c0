#!/bin/bash
TAG=${1:-s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_filters.py tests/test_slam.py tests/test_cpp_dropin.py tests/test_slam_replay.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --slam-record 8 --steps 1 --warmup 0 > $OUT/bench_slam100k.json 2> $OUT/bench.err
PGSLAM_HOST_INPUT_STAGE=1 python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --slam-record 8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_slam100k_hoststage.json 2>> $OUT/bench.err
python bench.py --workload slam --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_slam10k.json 2>> $OUT/bench.err
tail -3 $OUT/gputest.log
