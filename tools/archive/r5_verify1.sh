#!/bin/bash
# after the bounce buffer: cold slam_run passes (were 10-100 pauses of 25-40 ms each), the full GPU suite, the loop-closure leg with the shard proxy
TAG=${1:-v1}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for k in 1 2 3 4; do
  sleep 3
  ./tools/slam_run $SEQ --filters sensor --passes 1 > $OUT/cold_$k.json 2>> $OUT/err.log
  python3 -c "import json; d=json.loads(open('$OUT/cold_$k.json').read().strip().splitlines()[-1]); print('cold $k', d['pass_slam_s'], d['localizer_host_s'], d['icp_call_s']['p99'], d['icp_call_s']['max'])"
done
./tools/slam_run $SEQ --filters sensor --passes 4 > $OUT/warm.json 2>> $OUT/err.log
python3 -c "import json; d=json.loads(open('$OUT/warm.json').read().strip().splitlines()[-1]); print('warm', d['pass_slam_s'], d['localizer_host_s'])"
python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; tail -3 $OUT/gputest.log
python bench.py --workload loopclosure --steps 3 --warmup 1 --no-cpu-baseline --shard-proxy > $OUT/bench_lc.json 2>> $OUT/err.log
python3 -c "
import json; d=json.loads(open('$OUT/bench_lc.json').read().strip().splitlines()[-1]); print('lc', d['value'], d['roofline']['frac']); print(json.dumps(d['shard_proxy'], indent=0)[:1500])"
