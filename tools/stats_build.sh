#!/bin/bash
# diagnostics build (candidate counters) into the in-tree library path of THIS checkout; used on the GPU box only
set -e
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc ${STATS:--DPGICP_KNN_STATS} $EXTRA"
hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_stats.o
hipcc $F -c -x hip pgslam_amd/csrc/pgicp_api.cpp -o /tmp/a_stats.o
hipcc $F -c -x hip pgslam_amd/csrc/pgicp_comm.cpp -o /tmp/c_stats.o
hipcc --offload-arch=gfx950 -shared -fPIC -o pgslam_amd/lib/libpgicp.so /tmp/k_stats.o /tmp/a_stats.o /tmp/c_stats.o -ldl
