#!/bin/bash
# Timing experiment: the unseeded first pass run a second time with perfect seeds (what better first bounds could be worth at most)
set -u
OUT=gpurun_out/reseed0
mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc -DPGICP_EXP_RESEED0"
hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_rs.o && hipcc $F -c pgslam_amd/csrc/pgicp_api.cpp -o /tmp/a_rs.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_rs.so /tmp/k_rs.o /tmp/a_rs.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread || exit 1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
export PGICP_LIB_OVERRIDE=/tmp/lib_rs.so
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_h -o t -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-fixed30 --no-cpu-baseline --no-host-input --no-workloads --no-profile > $REPO/$OUT/trace_h.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace_lc -o t -- python3 $REPO/bench.py --workload loopclosure --pairs 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $REPO/$OUT/trace_lc.log 2>&1
cd $REPO
python3 - <<'P' > $OUT/result.txt
import csv, glob
for name in ('trace_h', 'trace_lc'):
    f = glob.glob('gpurun_out/reseed0/%s/**/*kernel_trace.csv' % name, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'k_knn_grid' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    print(name, 'k_knn_grid launches (us), last 14:', [round(x) for x in d[-14:]])
P
find $OUT -name '*.csv' -delete; rm -rf $OUT/trace_h $OUT/trace_lc
cat $OUT/result.txt; tail -3 $OUT/trace_h.log
