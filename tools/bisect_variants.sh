#!/bin/bash
# one failing stress case against compile-time variants of the kernels: tools/bisect_variants.sh SEED CASE "name:flags" ...
S=$1; C=$2; MAXC=${MAXC:-1073741824}; shift 2
run() { PGICP_LIB_OVERRIDE=$1 python3 tools/stress_parity.py 600 $S $C $MAXC 2>&1 | grep -E "MISMATCH|AssertionError|all equal" | head -3; }
echo "== intree"; run ""
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $f -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$n.so /tmp/k_$n.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
  echo "== $n ($f)"; run /tmp/lib_$n.so
done
