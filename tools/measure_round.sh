#!/bin/bash
# The round's measurement record (run on the GPU box through gpurun): default bench line, the kernel trace of the same
# workload, the HBM-traffic counters in their own passes, and the other workloads.  Output: gpurun_out/measure/ (the
# summaries are copied by hand into profiles/, named per round).
set -u
OUT=gpurun_out/measure
mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
# run NAME ARGS...: the printed line -> $OUT/NAME.json (what the driver parses), the full record bench.py wrote -> $OUT/NAME_full.json
run() { n=$1; shift; rm -f bench_full.json; python3 bench.py "$@" 2>/dev/null | tail -1 > $OUT/$n.json; cp bench_full.json $OUT/${n}_full.json 2>/dev/null; }
run bench_n1 --steps 20 --warmup 5     # (the driver's command line)
run bench_loopclosure --workload loopclosure --pairs 512 --steps 2 --warmup 1
run bench_stream_1 --workload stream --streams 1 --steps 2 --warmup 1
run bench_stream_4 --workload stream --streams 4 --steps 2 --warmup 1
run bench_stream_fleet16 --workload stream --streams 16 --fleet --steps 2 --warmup 1
run bench_slam --workload slam --steps 1 --warmup 0
run bench_slam100k --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --slam-record 8 --steps 1 --warmup 0
run bench_f64 --workload f64 --steps 5 --warmup 2
./tools/slam_run /tmp/pgslam_amd_seq_4500_10000_0.8.bin --mt --passes 3 > $OUT/slam_mt.json 2>/dev/null
./tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --mt --passes 4 > $OUT/slam100k_mt.json 2>/dev/null
run bench_loopclosure_shard_proxy --workload loopclosure --pairs 512 --steps 3 --warmup 1 --no-cpu-baseline --shard-proxy
python3 tools/bench_normals.py 2>/dev/null | grep -v amdgpu > $OUT/bench_normals.json
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
# kernel trace of the headline command (without the companion figures, so that it holds only the metric's launches)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -o t -- python3 $REPO/bench.py --no-fixed30 --no-cpu-baseline --no-host-input --no-workloads > $REPO/$OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_lc -o t -- python3 $REPO/bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $REPO/$OUT/trace_lc.log 2>&1
# the copy / compute overlap of the host-input pipeline: kernels and memory copies on one time line
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/$OUT/trace_host -o t -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-fixed30 --no-cpu-baseline --no-profile --no-workloads > $REPO/$OUT/trace_host.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_stream -o t -- python3 $REPO/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-input > $REPO/$OUT/trace_stream.log 2>&1
# the facade at sensor size: two passes of the 600-scan drive (the program itself after `--`)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_slam100k -o t -- $REPO/tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --passes 2 > $REPO/$OUT/trace_slam100k.log 2>&1
cd $REPO
python3 tools/trace_summary.py $OUT/trace > $OUT/trace_summary.txt 2>&1
python3 tools/trace_summary.py $OUT/trace_lc > $OUT/trace_lc_summary.txt 2>&1
python3 tools/overlap_summary.py $OUT/trace_host > $OUT/host_input_overlap.txt 2>&1
python3 tools/trace_summary.py $OUT/trace_stream > $OUT/trace_stream_summary.txt 2>&1
python3 tools/timeline.py $OUT/trace_stream > $OUT/stream_timeline_last_scan.txt 2>&1
python3 tools/trace_summary.py $OUT/trace_slam100k > $OUT/trace_slam100k_summary.txt 2>&1
rm -rf $OUT/trace/*.db $OUT/trace_lc/*.db $OUT/trace_host/*.db $OUT/trace_stream/*.db $OUT/trace_slam100k/*.db 2>/dev/null
# (the raw per-dispatch traces are tens of MB: only the summaries and the per-kernel statistics travel back)
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_memory_copy_trace.csv' -delete; find $OUT -name '*_agent_info.csv' -delete
# counters: every --pmc pass its own run (tools/pmc_round.sh) -> knn_traffic*.json, knn_pmc.json
bash tools/pmc_round.sh measure/pmc > $OUT/pmc.log 2>&1
for f in $OUT/bench_*.json $OUT/slam_mt.json; do echo "$f: $(cut -c1-300 $f)"; done
tail -6 $OUT/pmc.log | cut -c1-300; head -8 $OUT/trace_summary.txt; cat $OUT/host_input_overlap.txt
