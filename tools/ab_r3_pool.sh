#!/bin/bash
# round 3: the pooled fast matcher (k_knn_pool) against k_knn_grid on the headline; variants built on the box
mkdir -p gpurun_out/r3
python3 bench.py --prepare-only > /dev/null 2>&1
variant() {  # name, extra flags
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $2"
  hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$1.so /tmp/k_$1.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
}
{
tools/ab_headline.sh "PGICP_POOL=0" "PGICP_POOL=1" || tail -5 /tmp/ab_err.txt
for v in "q128:-DPGICP_POOL_Q=128 -DPGICP_POOL_RANGES=512" "q192:-DPGICP_POOL_Q=192 -DPGICP_POOL_RANGES=640" "q256p4:-DPGICP_POOL_PAIRS=4" "q512:-DPGICP_POOL_Q=512 -DPGICP_POOL_RANGES=1280"; do
  n=${v%%:*}; f=${v#*:}
  variant $n "$f" && tools/ab_headline.sh "PGICP_LIB_OVERRIDE=/tmp/lib_$n.so PGICP_POOL=1"
done
} 2>&1 | tee gpurun_out/r3/ab_pool.txt
tail -3 /tmp/ab_err.txt
