#!/bin/bash
# round 3: the C++ facade after the lazy-context / prefetch changes: C++ GPU tests, SLAM bench leg, ST vs MT runs with revisit counts
mkdir -p gpurun_out/r3
timeout 1500 python3 -m pytest tests/test_cpp_dropin.py tests/test_slam.py tests/test_slam_replay.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/r3/tests_cpp.txt
python3 bench.py --workload slam --steps 1 --warmup 0 2>gpurun_out/r3/slam.err | tail -1 > gpurun_out/r3/bench_slam.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench_slam.json"))
print("ST", d["value"], {k:d["slam"].get(k) for k in ("keyframes","loops_closed","loop_candidates_tried","keyframes_revisiting_within_3m_by_truth","keyframes_revisiting_within_3m_by_estimate","tracking_error_rms_m")}, d["replay_vs_oracle"], (d["roofline"] or {}).get("frac"))
PY
SEQ=$(ls /tmp/pgslam_amd_seq_4500_10000_0.8.bin)
for i in 1 2; do tools/slam_run $SEQ --mt | tail -1 | tee -a gpurun_out/r3/slam_mt.json; done
