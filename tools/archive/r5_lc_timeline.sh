#!/bin/bash
# Per-launch durations of one loop-closure step (512 pairs): which launches of the matcher carry the step
set -u
OUT=gpurun_out/lc_timeline
mkdir -p $OUT
python3 bench.py --workload loopclosure --prepare-only > /dev/null 2>&1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace -o t -- python3 $REPO/bench.py --workload loopclosure --pairs 512 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $REPO/$OUT/trace.log 2>&1
cd $REPO
python3 - <<'P' > $OUT/timeline.txt
import csv, glob
f = glob.glob('gpurun_out/lc_timeline/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: from the last k_mbin on
last = max(i for i, r in enumerate(rows) if 'k_centroid_bbox_b' in r['Kernel_Name'])
t0 = int(rows[last]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in rows[last:]:
    n = r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev_end) / 1e3
    if gap > 30: print(f"          -- idle {gap:8.1f} us --")
    prev_end = max(prev_end, int(r['End_Timestamp']))
    busy += d
    if d < 20: continue
    print(f"{(int(r['Start_Timestamp'])-t0)/1e6:9.3f} ms  {n:24s} {d:9.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']}")
print(f"step: {(prev_end - t0)/1e6:.3f} ms from the first build kernel to the last kernel, kernels busy {busy/1e3:.3f} ms")
P
find $OUT -name '*.csv' -delete; rm -rf $OUT/trace
cat $OUT/timeline.txt
