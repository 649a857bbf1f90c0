python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
for i in 1 2 3; do
python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
done
python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --sync-rebuild 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --workload stream --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
