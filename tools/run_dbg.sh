python3 bench.py --prepare-only >/dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scan2map', d['value'])"
for i in 1 2; do python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160; done
python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c80-140
