#!/bin/bash
# Three counter passes (SQ instruction mix, SQ wave time, TA/TCP) for one fast-matcher variant:
#   tools/pmc_quick.sh kernel-name-substring PGICP_FAST_KERNEL-value outdir
KNAME=$1; export PGICP_FAST_KERNEL=$2; OUT=$3
mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $REPO/$OUT/p$i -o p -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-fixed30 > $REPO/$OUT/p$i.log 2>&1
done
cd $REPO
for k in 1 2 3; do python3 tools/pmc_summary.py $OUT/p$k $KNAME; done > $OUT/summary.txt
for k in 1 2 3; do python3 tools/pmc_summary.py $OUT/p$k $KNAME each | grep -v "^k_knn"; done > $OUT/per_dispatch.txt
rm -rf $OUT/p*/*/*.db 2>/dev/null
cat $OUT/summary.txt
