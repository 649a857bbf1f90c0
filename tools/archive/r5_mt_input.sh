#!/bin/bash
# MT flavour with and without the input-stage thread, warm passes, alternating; both sequences
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --prepare-only > /dev/null 2>&1
mkdir -p gpurun_out/mt_input
for k in 1 2 3; do
 for v in 0 1; do
  PGSLAM_MT_INPUT_THREAD=$v ./tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --mt --passes 4 2>/dev/null > gpurun_out/mt_input/big_${v}_$k.json
  python3 -c "
import json; d=json.load(open('gpurun_out/mt_input/big_${v}_$k.json')); print('100k input_thread=$v', d['scans_per_s'], d.get('pass_wall_s'), d['input_stage_thread_s'], d['localizer_thread_s']['waiting_for_input_stage'])"
 done
done
for k in 1 2; do
 for v in 0 1; do
  PGSLAM_MT_INPUT_THREAD=$v ./tools/slam_run /tmp/pgslam_amd_seq_4500_10000_0.8.bin --mt --passes 3 2>/dev/null > gpurun_out/mt_input/small_${v}_$k.json
  python3 -c "
import json; d=json.load(open('gpurun_out/mt_input/small_${v}_$k.json')); print('10k input_thread=$v', d['scans_per_s'], d.get('pass_wall_s'), d['input_stage_thread_s'], d['localizer_thread_s']['waiting_for_input_stage'], d['keyframes'], d['loops_closed'])"
 done
done
