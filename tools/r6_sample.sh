#!/bin/bash
# Round 6: the matcher pass without correspondences run coarse to fine (SamplePass, PGICP_SAMPLE_STEPS) against the single launch
# (PGICP_SAMPLE_STEPS=0), per leg; parity tests first.
OUT=gpurun_out/r6s; mkdir -p $OUT
{
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bit_exact.py tests/test_gpu_edge_cases.py tests/test_gpu_matcher_state.py tests/test_slam_replay.py -m gpu -x -q 2>&1 | tail -5
python3 bench.py --prepare-only > /dev/null 2>&1
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
python3 bench.py --workload stream --prepare-only > /dev/null 2>&1
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
val() { python3 -c "
import json; d=json.load(open('bench_full.json')); r=d.get('roofline') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms/step; knn pass us', r.get('avg_launch_us') and round(r['avg_launch_us'],1), 'unseeded', r.get('avg_unseeded_launch_us') and round(r['avg_unseeded_launch_us'],1), 'seeded', r.get('avg_seeded_launch_us') and round(r['avg_seeded_launch_us'],1), 'frac', r.get('frac') and round(r['frac'],4))"; }
for t in 0 64,8 8 16 32,4 128,16,4 64,8 0; do
  echo -n "headline, PGICP_SAMPLE_STEPS=$t: "; rm -f bench_full.json
  PGICP_SAMPLE_STEPS=$t python3 bench.py --no-cpu-baseline --no-fixed30 --no-host-input --no-workloads > /dev/null 2>/tmp/err.txt; val
done
for t in 0 64,8 8 0 64,8; do
  echo -n "loop closing, PGICP_SAMPLE_STEPS=$t: "; rm -f bench_full.json
  PGICP_SAMPLE_STEPS=$t python3 bench.py --workload loopclosure --pairs 512 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
done
for t in 0 64,8 8; do
  echo -n "stream (one vehicle), PGICP_SAMPLE_STEPS=$t: "; rm -f bench_full.json
  PGICP_SAMPLE_STEPS=$t python3 bench.py --workload stream --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-host-input > /dev/null 2>/tmp/err.txt; val
done
SEQ=/tmp/pgslam_amd_seq_600_100000_0.8.bin
for t in 0 64,8 8 0 64,8 8; do
  echo -n "facade at sensor size, single thread, PGICP_SAMPLE_STEPS=$t: "; PGICP_SAMPLE_STEPS=$t ./tools/slam_run $SEQ --filters sensor --passes 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('scans_per_s','keyframes','loops_closed','map_rebuilds','overlap_probes_seeded','tracking_error_rms_m')}, d['localizer_host_s']['after_icp_parts']['overlap_probe'])"
done
echo -n "f64, default: "; rm -f bench_full.json; python3 bench.py --workload f64 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
echo -n "f64, PGICP_SAMPLE_STEPS=0: "; rm -f bench_full.json; PGICP_SAMPLE_STEPS=0 python3 bench.py --workload f64 --no-cpu-baseline > /dev/null 2>/tmp/err.txt; val
} 2>&1 | tee $OUT/sample_steps.txt
