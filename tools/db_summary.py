#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run kept as a rocpd database: db_summary.py trace_results.db [rows]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for n, cnt, t, mx in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    nm = n.split('(')[0].split('<')[0].split('::')[-1]
    print(f"{nm:28s} calls {cnt:5d} total {t/1e3:9.3f} ms avg {t/cnt:9.1f} us max {mx:9.1f} us  {100*t/tot:5.1f}%")
if len(sys.argv) > 3:
    # db_summary.py db rows pattern [count]: the last `count` dispatches whose kernel name matches the regular expression `pattern`, in launch order
    pat, cnt = sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    sel = [(n.split('(')[0].split('<')[0].split('::')[-1], s, e) for n, s, e in rows if re.search(pat, n)]
    t0 = sel[-cnt][1] if len(sel) >= cnt else (sel[0][1] if sel else 0)
    for n, s, e in sel[-cnt:]:
        print(f"  {n:16s} t={(s - t0) / 1e6:9.3f} ms dur {(e - s) / 1e3:9.1f} us")
