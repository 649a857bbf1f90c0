#!/bin/bash
# unit counters of the double fast matcher (two --pmc passes of the f64 leg) -> knn_pmc_f64.json
OUT=gpurun_out/${1:-r4pmcf64}; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp
F64="--workload f64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum --output-format csv -d $R/$OUT/grbm -o p -- python3 $R/bench.py $F64 > $R/$OUT/grbm.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/$OUT/sq -o p -- python3 $R/bench.py $F64 > $R/$OUT/sq.log 2>&1
cd $R
python3 tools/pmc_valu.py $OUT/grbm $OUT/sq $OUT/knn_pmc_f64.json 12800000 "k_knn_grid<double" 6.4921875 | cut -c1-600
rm -rf $OUT/*/*/*.db 2>/dev/null
