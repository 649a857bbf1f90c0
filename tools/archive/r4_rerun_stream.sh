#!/bin/bash
# the streaming parts of tools/measure_round.sh again (into the same gpurun_out/measure/)
OUT=gpurun_out/measure; mkdir -p $OUT/pmc
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_n1.json
python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_stream_1.json
python3 bench.py --workload stream --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_stream_4.json
python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_stream_fleet16.json
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_stream -o t -- python3 $REPO/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-input > $REPO/$OUT/trace_stream.log 2>&1
ST="--workload stream --streams 1 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-host-input"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/pmc/st_fetch -o p -- python3 $REPO/bench.py $ST > $REPO/$OUT/pmc/st_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/$OUT/pmc/st_write -o p -- python3 $REPO/bench.py $ST > $REPO/$OUT/pmc/st_write.log 2>&1
cd $REPO
python3 tools/trace_summary.py $OUT/trace_stream > $OUT/trace_stream_summary.txt 2>&1
python3 tools/timeline.py $OUT/trace_stream > $OUT/stream_timeline_last_scan.txt 2>&1
python3 tools/pmc_traffic.py $OUT/pmc/st_fetch $OUT/pmc/st_write $OUT/pmc/knn_traffic_stream.json 100000 2000000 1 k_knn_grid stream
for d in st_fetch st_write; do python3 tools/pmc_summary.py $OUT/pmc/$d > $OUT/pmc/${d}_all_kernels.txt 2>&1; done
rm -rf $OUT/trace_stream/*.db $OUT/pmc/*/*/*.db 2>/dev/null
for f in $OUT/bench_stream_*.json $OUT/bench_n1.json; do echo "$f: $(cut -c1-200 $f)"; done
