#!/bin/bash
# round-4 quick check on the GPU box: parity subset, headline and loop-closure bench lines, kernel trace of both
# usage: tools/r4_quick.sh TAG [full]
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$2" = "full" ]; then
  python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
else
  python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_matcher_state.py tests/test_gpu_edge_cases.py tests/test_gpu_stress.py tests/test_gpu_knobs.py tests/test_gpu_full_size.py -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
fi
python bench.py --steps 10 --warmup 3 --no-workloads --no-cpu-baseline --no-host-input > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload loopclosure --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_lc.json 2>> $OUT/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-workloads --no-cpu-baseline --no-host-input --no-fixed30 --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_lc -o trace -- python3 $R/bench.py --workload loopclosure --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $R
python tools/db_summary.py $OUT/prof/trace_results.db 40 > $OUT/trace_summary.txt 2>&1
python tools/db_summary.py $OUT/prof_lc/trace_results.db 45 k_knn 42 > $OUT/trace_summary_lc.txt 2>&1
rm -rf $OUT/prof $OUT/prof_lc
tail -3 $OUT/gputest.log
