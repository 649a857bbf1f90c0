#!/bin/bash
# compile-time variants of the kernels on the f64 leg: tools/r4_f64_variants.sh "name:flags" ...
mkdir -p gpurun_out/r4f64v
python3 bench.py --prepare-only > /dev/null 2>&1
run() { echo -n "$1: "; env $2 python3 bench.py --workload f64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],2), 'ms; knn launch', round(d['roofline']['avg_launch_us'],1), 'us')"; }
{
run intree X=1
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $f"
  hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$n.so /tmp/k_$n.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread && run $n PGICP_LIB_OVERRIDE=/tmp/lib_$n.so
done
run intree X=1
} 2>&1 | tee gpurun_out/r4f64v/summary.txt
