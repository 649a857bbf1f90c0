#!/bin/bash
# Experiment: what a search cap in the UNSEEDED first pass is worth -- the cap taken from the previous call's exact threshold
# (selection hints), i.e. the best a sampled estimate could do.  tools/r5_cap0.sh
mkdir -p gpurun_out/cap0
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
variant() {
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $2"
  hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$1.so /tmp/k_$1.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
}
run() {
  echo -n "$1 headline: "; PGICP_LIB_OVERRIDE=$2 python3 bench.py --steps 20 --warmup 5 --no-fixed30 --no-cpu-baseline --no-host-input --no-workloads 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['roofline'].get('kernel_ms'))"
  echo -n "$1 loopclosure: "; PGICP_LIB_OVERRIDE=$2 python3 bench.py --workload loopclosure --pairs 512 --steps 3 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
}
{
run intree ""
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  variant $n "$f" && run $n /tmp/lib_$n.so
done
run intree ""
} 2>&1 | tee gpurun_out/cap0/result.txt
