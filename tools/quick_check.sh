#!/bin/bash
# quick A/B on the GPU box: GPU suite, then the value of each workload's bench line (gpurun_out/qc/)
mkdir -p gpurun_out/qc
if [ "$1" != "notest" ]; then python -m pytest tests -x -q -m gpu > gpurun_out/qc/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/qc/pytest.txt | tail -3; fi
val() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['value'],1), d.get('ms_per_step') and round(d['ms_per_step'],2), d.get('mean_iterations'), d.get('ms_per_scan_per_vehicle'))" $1; }
python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/qc/stream1.json; val gpurun_out/qc/stream1.json
python3 bench.py --workload stream --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/qc/stream1b.json; val gpurun_out/qc/stream1b.json
python3 bench.py --workload stream --streams 16 --fleet --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/qc/fleet16.json; val gpurun_out/qc/fleet16.json
python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > gpurun_out/qc/lc.json; val gpurun_out/qc/lc.json
python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-profile 2>/dev/null | tail -1 > gpurun_out/qc/n1.json; val gpurun_out/qc/n1.json
if [ "$2" == "slam" ]; then python3 bench.py --workload slam --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/qc/slam.json; val gpurun_out/qc/slam.json; fi
