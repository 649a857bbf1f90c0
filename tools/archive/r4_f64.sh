mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_residual" > gpurun_out/r4k/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4k/gputest.log
python bench.py --workload f64 --steps 3 --warmup 1 > gpurun_out/r4k/bench_f64.json 2> gpurun_out/r4k/bench.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4k/prof -o trace -- python3 $R/bench.py --workload f64 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; python tools/db_summary.py gpurun_out/r4k/prof/trace_results.db 30 k_knn 24 > gpurun_out/r4k/trace_f64.txt 2>&1; rm -rf gpurun_out/r4k/prof
tail -3 gpurun_out/r4k/gputest.log
