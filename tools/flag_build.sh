#!/bin/bash
# experiment: rebuild the library in this checkout with extra compiler flags ($EXTRA) -- GPU box only
set -e
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $EXTRA"
hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_x.o
hipcc $F -c -x hip pgslam_amd/csrc/pgicp_api.cpp -o /tmp/a_x.o
hipcc --offload-arch=gfx950 -shared -fPIC -o pgslam_amd/lib/libpgicp.so /tmp/k_x.o /tmp/a_x.o
