python3 bench.py --prepare-only >/dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-120
python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --workload stream --streams 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-160
python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c80-140
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/strsync -o s -- python3 $R/bench.py --workload stream --streams 1 --steps 1 --warmup 1 --sync-rebuild > $R/gpurun_out/strsync.log 2>&1)
python3 tools/timeline.py gpurun_out/strsync | tail -42
