#!/bin/bash
# A/B of compile-time variants of the kernels on the headline (+ stream): tools/ab_variants.sh "name:flags" ...
mkdir -p gpurun_out/r3
python3 bench.py --prepare-only > /dev/null 2>&1
variant() {
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $2"
  hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$1.so /tmp/k_$1.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
}
{
tools/ab_headline.sh "X=intree"
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  variant $n "$f" && tools/ab_headline.sh "PGICP_LIB_OVERRIDE=/tmp/lib_$n.so" && echo -n "   stream: " && PGICP_LIB_OVERRIDE=/tmp/lib_$n.so python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
done
tools/ab_headline.sh "X=intree"
} 2>&1 | tee gpurun_out/r3/ab_variants.txt
