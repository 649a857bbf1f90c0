#!/bin/bash
# unseeded vs seeded launch time of the fast matcher against the grid's cell size (is a second, finer index for the unseeded pass worth it?)
OUT=gpurun_out/${1:-r4cs}; mkdir -p $OUT
python3 bench.py --prepare-only > /dev/null 2>&1
for kx in 4 2 8; do
for s in 0.35 0.5 0.71 1.0 1.41; do
  PGICP_KX=$kx PGICP_CELL_SCALE=$s python3 bench.py --steps 4 --warmup 2 --no-workloads --no-cpu-baseline --no-host-input --no-fixed30 2>/dev/null | tail -1 > $OUT/b_${kx}_$s.json
  python3 - $OUT/b_${kx}_$s.json $kx $s <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
except Exception as e:
    print("kx %s scale %s: failed (%s)" % (sys.argv[2], sys.argv[3], type(e).__name__)); sys.exit(0)
print("kx %s scale %s: %.0f scans/s  %.2f ms/step  unseeded %.0f us  seeded %.0f us  avg %.0f us  set_map %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"],
      r.get("avg_unseeded_launch_us", 0), r.get("avg_seeded_launch_us", 0), r["avg_launch_us"], d.get("set_map_ms")))
PY
done; done | tee $OUT/summary.txt
