#!/bin/bash
TAG=${1:-fb}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests/test_slam.py tests/test_cpp_dropin.py tests/test_slam_replay.py -m gpu -x -q 2>&1 | tail -15
python bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python bench.py --workload slam --prepare-only > /dev/null 2>&1
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
./tools/slam_run $SEQ --filters sensor --passes 4 > $OUT/st.json 2>> $OUT/err.log
python3 -c "import json; d=json.loads(open('$OUT/st.json').read().strip().splitlines()[-1]); print('ST 100k', d['pass_slam_s'], json.dumps(d['localizer_host_s']), 'kf', d['keyframes'], 'loops', d['loops_closed'], 'dev cands', d['loop_candidates_assembled_on_device'], 'resident', d['keyframes_resident'])"
for k in 1 2 3; do ./tools/slam_run $SEQ --filters sensor --mt > $OUT/mt_$k.json 2>> $OUT/err.log; python3 -c "import json; d=json.loads(open('$OUT/mt_$k.json').read().strip().splitlines()[-1]); print('MT 100k', d['scans_per_s'], 'loops', d['loops_closed'], 'batches', d['loop_batches'], 'largest', d['largest_loop_batch'], 'dev', d['loop_batches_on_device'], 'err', d['tracking_error_last_m'])"; done
SEQ2=/tmp/pgslam_amd_seq_4500_10000_0.8.bin
./tools/slam_run $SEQ2 --passes 2 > $OUT/st10k.json 2>> $OUT/err.log
python3 -c "import json; d=json.loads(open('$OUT/st10k.json').read().strip().splitlines()[-1]); print('ST 10k', d['pass_slam_s'], d['scans_per_s'], 'loops', d['loops_closed'], 'rms', d['tracking_error_rms_m'])"
./tools/slam_run $SEQ2 --mt > $OUT/mt10k.json 2>> $OUT/err.log; python3 -c "import json; d=json.loads(open('$OUT/mt10k.json').read().strip().splitlines()[-1]); print('MT 10k', d['scans_per_s'], 'loops', d['loops_closed'], 'batches', d['loop_batches'], 'largest', d['largest_loop_batch'])"
tail -5 $OUT/err.log
