# per-launch durations of the matcher kernels from a rocprofv3 kernel trace, in launch order
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    tag = 'G' if 'k_knn_grid' in n else 'M' if 'k_knn_med' in n else 'S' if 'k_knn_slow' in n else None
    if tag:
        out.append(f"{tag}{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}")
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print(' '.join(out[-last:]))
