#!/bin/bash
# Round 6: super-cells opened per trip of the wave-per-query path's walk (PGICP_SLOW_OPEN; 1 = round 5), per leg; parity tests first
OUT=gpurun_out/r6so; mkdir -p $OUT
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
{
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bit_exact.py tests/test_gpu_edge_cases.py tests/test_gpu_matcher_state.py tests/test_local_mapper.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | grep -v '^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl' | tail -5
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
for k in 1 2 3 4; do
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc -DPGICP_SLOW_OPEN=$k"
  mkdir -p /tmp/libk$k
  hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$k.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libk$k/libpgicp.so /tmp/k_$k.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
done
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step')"; }
for rep in 1 2; do for k in 1 2 3 4; do
  export PGICP_LIB_OVERRIDE=/tmp/libk$k/libpgicp.so
  echo -n "stream 1, open $k: "; rm -f bench_full.json; python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "stream fleet 16, open $k: "; rm -f bench_full.json; python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
  echo -n "loop closing, open $k: "; rm -f bench_full.json; python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
  echo -n "headline, open $k: "; rm -f bench_full.json; python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/err.txt; val
  unset PGICP_LIB_OVERRIDE
  echo -n "facade 100k, open $k: "; LD_LIBRARY_PATH=/tmp/libk$k:$LD_LIBRARY_PATH ./tools/slam_run $SEQ --filters sensor --passes 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scans_per_s'], d['keyframes'], d['loops_closed'])"
done; done
} 2>&1 | tee $OUT/slow_open.txt
cd /tmp && export TMPDIR=/tmp
for k in 1 3; do
  PGICP_LIB_OVERRIDE=/tmp/libk$k/libpgicp.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/trace_$k -o t -- python3 $OLDPWD/bench.py --workload stream --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-input > $OLDPWD/$OUT/trace_$k.log 2>&1
  python3 $OLDPWD/tools/trace_summary.py $OLDPWD/$OUT/trace_$k | head -6 > $OLDPWD/$OUT/trace_${k}_summary.txt 2>&1
  rm -rf $OLDPWD/$OUT/trace_$k
done
cd $OLDPWD; cat $OUT/trace_1_summary.txt $OUT/trace_3_summary.txt
