#!/bin/bash
# A/B of the in-tree library against pgslam_amd/lib/libpgicp_base.so (a build of an earlier commit) on the headline, then the GPU suite
mkdir -p gpurun_out/r3
python3 bench.py --prepare-only > /dev/null 2>&1
B=$PWD/pgslam_amd/lib/libpgicp_base.so
tools/ab_headline.sh "PGICP_LIB_OVERRIDE=$B" "X=1" "PGICP_LIB_OVERRIDE=$B" "X=1" 2>&1 | tee gpurun_out/r3/ab_lib.txt
tail -3 /tmp/ab_err.txt
for w in "--workload stream --steps 2 --warmup 1 --no-cpu-baseline --no-host-input" "--workload loopclosure --steps 2 --warmup 1 --no-cpu-baseline --no-profile"; do
  for lib in "$B" ""; do
    echo -n "$w lib=${lib:-new}: "; PGICP_LIB_OVERRIDE=$lib python3 bench.py $w 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],1))"
  done
done 2>&1 | tee -a gpurun_out/r3/ab_lib.txt
if [ "$1" = "tests" ]; then
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -12 | tee gpurun_out/r3/tests_all.txt
fi
