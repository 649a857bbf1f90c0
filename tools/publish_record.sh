#!/bin/bash
# copies the files of a tools/measure_round.sh run (gpurun_out/measure/) into profiles/ under this round's names:
# tools/publish_record.sh r03
R=${1:?round prefix, e.g. r03}; M=gpurun_out/measure; P=profiles
cp $M/bench_n1.json $P/${R}_bench_n1.json; cp $M/bench_n1_full.json $P/${R}_bench_n1_full.json
cp $M/bench_loopclosure_full.json $P/${R}_bench_loopclosure.json
cp $M/bench_normals.json $P/${R}_bench_normals.json
cp $M/bench_slam_full.json $P/${R}_bench_slam.json
cp $M/bench_stream_1_full.json $P/${R}_bench_stream_1vehicle.json
cp $M/bench_stream_4_full.json $P/${R}_bench_stream_4vehicles.json
cp $M/bench_stream_fleet16_full.json $P/${R}_bench_stream_fleet16.json
cp $M/host_input_overlap.txt $P/${R}_host_input_overlap.txt
cp $M/trace_lc/t_kernel_stats.csv $P/${R}_loopclosure_kernel_stats.csv
cp $M/trace_lc_summary.txt $P/${R}_loopclosure_trace_summary.txt
cp $M/trace/t_kernel_stats.csv $P/${R}_rocprofv3_kernel_stats.csv
cp $M/trace_summary.txt $P/${R}_trace_summary.txt
cp $M/slam_mt.json $P/${R}_slam_run_mt.json
cp $M/bench_slam100k_full.json $P/${R}_bench_slam100k.json
cp $M/bench_f64_full.json $P/${R}_bench_f64.json
cp $M/slam100k_mt.json $P/${R}_slam100k_run_mt.json 2>/dev/null
cp $M/bench_loopclosure_shard_proxy_full.json $P/${R}_bench_loopclosure_shard_proxy.json 2>/dev/null
cp $M/trace_slam100k_summary.txt $P/${R}_slam100k_trace_summary.txt 2>/dev/null
cp $M/trace_stream_summary.txt $P/${R}_stream_kernel_totals.txt
cp $M/stream_timeline_last_scan.txt $P/${R}_stream_timeline_last_scan.txt
# (the counter records both under their standing names -- bench.py reads profiles/knn_traffic.json / knn_pmc.json -- and under the round's, each with its commit inside)
for f in knn_traffic knn_pmc knn_traffic_loopclosure knn_traffic_stream knn_traffic_f64 knn_traffic_slam; do cp $M/pmc/$f.json $P/$f.json; cp $M/pmc/$f.json $P/${R}_$f.json; done
mkdir -p $P/${R}_pmc; cp $M/pmc/*_all_kernels.txt $M/pmc/head_*_per_dispatch.txt $M/pmc/digest.log $P/${R}_pmc/
python3 tools/measure_digest.py $M > $P/${R}_digest.txt 2>&1
