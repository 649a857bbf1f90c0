#!/bin/bash
# the MT flavour of the facade at sensor size, now: three runs of each flavour
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
mkdir -p gpurun_out/mt_now
for k in 1 2; do
  ./tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --mt --passes 4 2>/dev/null > gpurun_out/mt_now/mt_$k.json
  python3 -c "
import json; d=json.load(open('gpurun_out/mt_now/mt_$k.json')); print('mt', d['scans_per_s'], d['wall_s'], d.get('pass_wall_s'), {k:v for k,v in d.items() if 'thread' in k or 'loop' in k or 'rebuild' in k})"
done
./tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --passes 3 2>/dev/null > gpurun_out/mt_now/st.json
python3 -c "
import json; d=json.load(open('gpurun_out/mt_now/st.json')); print('st', d['scans_per_s'], d.get('pass_slam_s'), d.get('localizer_host_s'))"
PGSLAM_MT_INPUT_THREAD=1 ./tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --mt --passes 4 2>/dev/null > gpurun_out/mt_now/mt_input.json
python3 -c "
import json; d=json.load(open('gpurun_out/mt_now/mt_input.json')); print('mt+input thread', d['scans_per_s'], d.get('pass_wall_s'), d['input_stage_thread_s'], d['localizer_thread_s'])"
