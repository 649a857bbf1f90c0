#!/bin/bash
# stream / fleet / loop-closing values of compile-time variants, three runs each: tools/ab_stream_lib.sh "name:flags" ...
mkdir -p gpurun_out/r3
python3 bench.py --prepare-only > /dev/null 2>&1
val() { python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1), end=' ')"; }
for v in "intree:" "$@"; do
  n=${v%%:*}; f=${v#*:}; lib=""
  if [ "$n" != "intree" ]; then
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $f -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$n.so /tmp/k_$n.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
    lib=/tmp/lib_$n.so
  fi
  echo -n "$n stream: "; for i in 1 2 3; do PGICP_LIB_OVERRIDE=$lib python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | val; done; echo
  echo -n "$n fleet16: "; for i in 1 2; do PGICP_LIB_OVERRIDE=$lib python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | val; done; echo
done 2>&1 | tee gpurun_out/r3/ab_stream_lib.txt
