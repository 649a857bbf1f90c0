python3 bench.py --prepare-only >/dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
for i in 1 2; do
python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --workload stream --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
done
python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-120
