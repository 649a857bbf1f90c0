#!/bin/bash
# Counter passes for a fast matcher kernel (GPU box; each pass is its own rocprofv3 --pmc run).
#   tools/pmc_fast_kernel.sh [kernel-name-substring] [PGICP_FAST_KERNEL value] [out dir]
KNAME=${1:-k_knn_quad}
export PGICP_FAST_KERNEL=${2:-1}
OUT=${3:-gpurun_out/pmcf}
mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $REPO/$OUT/p$i -o p -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-fixed30 --no-host-input --no-workloads > $REPO/$OUT/p$i.log 2>&1
done
cd $REPO
for k in 1 2 3 4 5 6; do python3 tools/pmc_summary.py $OUT/p$k $KNAME; done > $OUT/summary.txt
python3 tools/pmc_summary.py $OUT/p3 $KNAME each > $OUT/per_dispatch_sq.txt
python3 tools/pmc_summary.py $OUT/p5 $KNAME each > $OUT/per_dispatch_tcp.txt
rm -rf $OUT/p*/*/*.db 2>/dev/null
cat $OUT/summary.txt
