#!/bin/bash
# The round's counter record (GPU box, through gpurun; PGSLAM_COMMIT in the environment is written into every json): every rocprofv3 --pmc pass is its own run, no trace domains with it.
#   headline: FETCH_SIZE | WRITE_SIZE | GRBM+TA | SQ instruction mix      -> knn_traffic.json, knn_pmc.json
#   loop closing, streaming, f64: FETCH_SIZE | WRITE_SIZE                    -> knn_traffic_<leg>.json
OUT=gpurun_out/${1:-r4pmc}; mkdir -p $OUT; rm -f $OUT/passes.log
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-fixed30 --no-host-input --no-workloads"
LC="--workload loopclosure --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
ST="--workload stream --streams 1 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-host-input"
F64="--workload f64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
pass() { # name, counters, bench args
  local name=$1 ctr=$2; shift 2
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $R/$OUT/$name -o p -- python3 $R/bench.py "$@" > $R/$OUT/$name.log 2>&1
  echo "$name rc=$?" >> $R/$OUT/passes.log
}
pass head_fetch FETCH_SIZE $HEAD
pass head_write WRITE_SIZE $HEAD
pass head_grbm "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" $HEAD
pass head_sq "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" $HEAD
pass lc_fetch FETCH_SIZE $LC
pass lc_write WRITE_SIZE $LC
pass st_fetch FETCH_SIZE $ST
pass st_write WRITE_SIZE $ST
pass f64_fetch FETCH_SIZE $F64
pass f64_write WRITE_SIZE $F64
# the SLAM facade (C++ driver, first 1 500 scans of the configs[3] sequence; the program itself after `--`)
cd $R; python3 bench.py --workload slam --prepare-only --slam-scans 1500 > /dev/null 2>&1; python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=slam_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/$n -o p -- $R/tools/slam_run /tmp/pgslam_amd_seq_1500_10000_0.8.bin > $R/$OUT/$n.log 2>&1
  echo "$n rc=$?" >> $R/$OUT/passes.log
done
cd $R
python3 tools/pmc_traffic.py $OUT/head_fetch $OUT/head_write $OUT/knn_traffic.json 100000 1000000 128 > $OUT/digest.log 2>&1
python3 tools/pmc_valu.py $OUT/head_grbm $OUT/head_sq $OUT/knn_pmc.json 12800000 k_knn_grid 6.4921875 >> $OUT/digest.log 2>&1
python3 tools/pmc_traffic.py $OUT/lc_fetch $OUT/lc_write $OUT/knn_traffic_loopclosure.json 100000 100000 512 k_knn_grid loopclosure >> $OUT/digest.log 2>&1
python3 tools/pmc_traffic.py $OUT/st_fetch $OUT/st_write $OUT/knn_traffic_stream.json 100000 2000000 1 k_knn_grid stream >> $OUT/digest.log 2>&1
python3 tools/pmc_traffic.py $OUT/f64_fetch $OUT/f64_write $OUT/knn_traffic_f64.json 100000 1000000 128 k_knn_grid f64 >> $OUT/digest.log 2>&1
python3 tools/pmc_traffic.py $OUT/slam_fetch $OUT/slam_write $OUT/knn_traffic_slam.json 10000 30000 1 k_knn_grid slam >> $OUT/digest.log 2>&1
for d in head_fetch head_write lc_fetch lc_write st_fetch st_write f64_fetch f64_write; do python3 tools/pmc_summary.py $OUT/$d > $OUT/${d}_all_kernels.txt 2>&1; done
python3 tools/pmc_summary.py $OUT/head_grbm k_knn_grid each > $OUT/head_grbm_per_dispatch.txt 2>&1
python3 tools/pmc_summary.py $OUT/head_sq k_knn_grid each > $OUT/head_sq_per_dispatch.txt 2>&1
rm -rf $OUT/*/*/*.db 2>/dev/null
cat $OUT/passes.log; cat $OUT/digest.log | cut -c1-400
