#!/bin/bash
# Round 6, VERDICT item 2: compile-time variants of the fast matcher, each measured the same way on one box --
#   headline (two runs), loop closing, and SQ_INSTS_VALU of k_knn_grid over one headline step (its own rocprofv3 --pmc pass).
# tools/r6_variants.sh "name:flags" ...     -> gpurun_out/r6v/variants.txt
OUT=gpurun_out/r6v; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
R=$PWD
variant() {
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ipgslam_amd/csrc $2"
  hipcc $F -c -x hip pgslam_amd/csrc/kernels.hip -o /tmp/k_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$1.so /tmp/k_$1.o pgslam_amd/csrc/pgicp_api.o pgslam_amd/csrc/pgicp_comm.o -ldl -pthread
}
measure() { # name
  export PGICP_LIB_OVERRIDE=/tmp/lib_$1.so
  echo "#### $1"
  tools/ab_headline.sh "X=1" "X=2"
  echo -n "   loop closing: "; python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s, knn launch us', round(d['roofline']['avg_launch_us'],1))"
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/$OUT/pmc_$1 -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-fixed30 --no-host-input --no-workloads > $R/$OUT/pmc_$1.log 2>&1 )
  python3 - $OUT/pmc_$1 <<'P'
import csv, glob, sys, collections
tot = collections.Counter(); n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_knn_grid" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n += r["Counter_Name"] == "SQ_INSTS_VALU"
q = 12_800_000 * 6.4921875
print("   k_knn_grid over one step (%d launches): " % n + ", ".join("%s %.4g" % (k, v) for k, v in sorted(tot.items())) +
      "; VALU lane-ops per active query-iteration %.0f" % (64.0 * tot["SQ_INSTS_VALU"] / q))
P
  rm -rf $OUT/pmc_$1/*/*.db 2>/dev/null
  unset PGICP_LIB_OVERRIDE
}
{
for v in "$@"; do n=${v%%:*}; f=${v#*:}; variant $n "$f" && measure $n; done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tee $OUT/variants.txt
