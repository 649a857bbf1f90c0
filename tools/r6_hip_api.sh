#!/bin/bash
# host side of one drive of the facade: every HIP API call with its duration (rocprofv3 --hip-trace), summed by call name
OUT=gpurun_out/r6api; mkdir -p $OUT
python3 bench.py --workload slam --slam-scans 600 --slam-points 100000 --slam-filters sensor --prepare-only > /dev/null 2>&1
python3 -c "import bench; bench.build_slam_run()" > /dev/null 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-trace --output-format csv -d $R/$OUT/trace -o t -- $R/tools/slam_run /tmp/pgslam_amd_seq_600_100000_0.8.bin --filters sensor --limit 200 > $R/$OUT/trace.log 2>&1
cd $R
python3 - $OUT/trace <<'P' > $OUT/hip_api_summary.txt 2>&1
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*hip_api_trace.csv', recursive=True)[0]
tot = collections.Counter(); cnt = collections.Counter(); mx = collections.Counter()
rows = list(csv.DictReader(open(f)))
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
    n = r['Function']; tot[n] += d; cnt[n] += 1; mx[n] = max(mx[n], d)
span = (max(int(r['End_Timestamp']) for r in rows) - min(int(r['Start_Timestamp']) for r in rows)) / 1e6
print('calls %d, span %.1f ms' % (len(rows), span))
for n, t in tot.most_common(25): print('%-40s calls %7d total %9.1f us avg %8.2f us max %9.1f us' % (n, cnt[n], t, t / cnt[n], mx[n]))
P
rm -rf $OUT/trace
cat $OUT/hip_api_summary.txt
