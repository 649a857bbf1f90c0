python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
for rep in 1 2 3 4; do for s in X=0 PGICP_D2H_DIRECT=1; do
  rm -f bench_full.json; env $s python3 bench.py --workload loopclosure --pairs 512 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt
  python3 -c "
import json; d=json.load(open('bench_full.json')); print('$s', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
