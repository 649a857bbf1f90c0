#!/bin/bash
TAG=${1:-hi}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_matcher_state.py tests/test_gpu_edge_cases.py tests/test_local_mapper.py tests/test_gpu_stress.py tests/test_slam.py -m gpu -x -q 2>&1 | tail -4
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for h in 0 1 0 1; do
PGICP_SEL_HINTS=$h ./tools/slam_run $SEQ --filters sensor --passes 4 > $OUT/st_$h.json 2>> $OUT/err.log
python3 -c "import json; d=json.loads(open('$OUT/st_$h.json').read().strip().splitlines()[-1]); print('hints $h ST 100k', d['pass_slam_s'], json.dumps(d['localizer_host_s']), d['tracking_error_rms_m'], d['keyframes'])"
done
for h in 0 1; do PGICP_SEL_HINTS=$h python bench.py --workload stream --steps 3 --warmup 1 --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('hints $h stream', d['scans_per_s_each_pass'], d['selection_guess_misses_per_scan'], d['final_position_error_m'])"; done
for h in 0 1; do PGICP_SEL_HINTS=$h python bench.py --steps 10 --warmup 3 --no-workloads --no-cpu-baseline --no-host-input 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('hints $h headline', d['value'], d['selection_guess_misses_per_step'], d['mean_iterations'])"; done
