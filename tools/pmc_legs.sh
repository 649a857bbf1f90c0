#!/bin/bash
# VALU / address-path busy of the fast matcher in the loop-closure leg and in the f64 leg (every --pmc pass its own run)
OUT=gpurun_out/pmc_lc; mkdir -p $OUT; rm -f $OUT/passes.log
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
LC="--workload loopclosure --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
F64="--workload f64 --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
pass() { local name=$1 ctr=$2; shift 2
  timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $R/$OUT/$name -o p -- python3 $R/bench.py "$@" > $R/$OUT/$name.log 2>&1
  echo "$name rc=$?" >> $R/$OUT/passes.log; }
pass lc_grbm "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" $LC
pass lc_sq "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" $LC
pass f64_grbm "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" $F64
pass f64_sq "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" $F64
cd $R
python3 tools/pmc_valu.py $OUT/lc_grbm $OUT/lc_sq $OUT/knn_pmc_loopclosure.json 51200000 k_knn_grid 5.5 > $OUT/digest.log 2>&1
python3 tools/pmc_valu.py $OUT/f64_grbm $OUT/f64_sq $OUT/knn_pmc_f64.json 12800000 k_knn_grid 6.4921875 >> $OUT/digest.log 2>&1
rm -rf $OUT/*/*/*.db 2>/dev/null; find $OUT -name '*counter_collection.csv' -size +8M -delete
cat $OUT/passes.log; cut -c1-600 $OUT/digest.log
python3 - <<'P'
import json
for n in ('knn_pmc_loopclosure', 'knn_pmc_f64'):
    try:
        d = json.load(open('gpurun_out/pmc_lc/%s.json' % n))
        print(n, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k != 'per_launch'})
        for l in d.get('per_launch', [])[:8]: print('   ', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in l.items()})
    except Exception as e: print(n, e)
P
