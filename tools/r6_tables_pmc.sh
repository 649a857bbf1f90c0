#!/bin/bash
# Round 6, item 3: fabric traffic of the fast matcher and the index-build kernels of the loop-closure leg, dense against succinct tables
OUT=gpurun_out/r6tp; mkdir -p $OUT
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
R=$PWD
LC="--workload loopclosure --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
cd /tmp && export TMPDIR=/tmp
for t in dense succinct; do
  export PGICP_TABLES=$t
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/${t}_fetch -o p -- python3 $R/bench.py $LC > $R/$OUT/${t}_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/${t}_write -o p -- python3 $R/bench.py $LC > $R/$OUT/${t}_write.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${t}_trace -o t -- python3 $R/bench.py $LC > $R/$OUT/${t}_trace.log 2>&1
done
unset PGICP_TABLES
cd $R
for t in dense succinct; do
  python3 tools/pmc_traffic.py $OUT/${t}_fetch $OUT/${t}_write $OUT/knn_traffic_loopclosure_$t.json 100000 100000 512 k_knn_grid loopclosure 2>&1 | cut -c1-300
  echo "== $t: kernels of one step (ms)"; python3 tools/trace_summary.py $OUT/${t}_trace 2>/dev/null | head -14
done 2>&1 | tee $OUT/tables_pmc.txt
rm -rf $OUT/*/*/*.db 2>/dev/null; find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +4M -delete
