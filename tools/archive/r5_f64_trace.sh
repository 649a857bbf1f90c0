#!/bin/bash
# kernel totals of the f64 leg (one step) next to the f32 headline's
set -u
OUT=gpurun_out/f64_trace; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/t64 -o t -- python3 $REPO/bench.py --workload f64 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $REPO/$OUT/t64.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/t32 -o t -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-fixed30 --no-cpu-baseline --no-host-input --no-workloads --no-profile > $REPO/$OUT/t32.log 2>&1
cd $REPO
python3 tools/trace_summary.py $OUT/t64 > $OUT/f64_summary.txt 2>&1
python3 tools/trace_summary.py $OUT/t32 > $OUT/f32_summary.txt 2>&1
rm -rf $OUT/t64 $OUT/t32
echo "== f64 (3 steps)"; head -14 $OUT/f64_summary.txt; echo "== f32 (3 steps)"; head -14 $OUT/f32_summary.txt
