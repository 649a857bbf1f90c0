"""Diagnostics build only (tools/stats_build.sh first): where the waves of the fast matcher kernel spend their time,
on the benchmark's batch.  PGICP_FAST_KERNEL=0 (the phase marks live in k_knn_grid)."""
import os, sys
os.environ["PGICP_KNN_STATS_DUMP"] = "1"
import numpy as np, torch
sys.path.insert(0, '.')
from bench import build_workload, CHAIN
from pgslam_amd import icp, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w = build_workload(100000, 1000000, 64)
dev = torch.device('cuda', 0)
rd = [torch.from_numpy(s).to(dev) for s in w.scans_xyz]
readings = [rd[b % 64] for b in range(B)]
T0 = [w.T_truth[b % 64] @ synth.perturbation(b) for b in range(B)]
ctx = icp.Context(0, **CHAIN)
mid = ctx.set_map(torch.from_numpy(w.map_xyz).to(dev), torch.from_numpy(w.map_nrm).to(dev))
ctx.align_batch(mid, readings, T0, raise_on_error=False)
ctx.debug_counters()                      # warm-up: counters reset
ctx.align_batch(mid, readings, T0, raise_on_error=False)
print(f'one step of {B} scans:', file=sys.stderr)
ctx.debug_counters()
ctx.close()
