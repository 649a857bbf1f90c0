"""Study (stats build + a two-line hook in k_knn_grid that presets best.d2 from d2_out when PGICP_PRESET=1, not kept in the
product): candidate counters of the unseeded pass with and without a perfect initial bound.  See DESIGN.md section 4, item 8."""
import sys, os, numpy as np, torch
sys.path.insert(0,'.')
from bench import build_workload, CHAIN
from pgslam_amd import icp
w=build_workload(100000,1000000,16)
dev=torch.device('cuda',0)
rd=[torch.from_numpy(s).to(dev) for s in w.scans_xyz]
B=16
ctx=icp.Context(0, **dict(CHAIN, max_iters=1, min_diff_rot=0.0, min_diff_trans=0.0))
mid=ctx.set_map(torch.from_numpy(w.map_xyz).to(dev), torch.from_numpy(w.map_nrm).to(dev))
ctx.align_batch(mid, rd[:B], w.T_init[:B], raise_on_error=False); ctx.debug_counters()
import time
for k in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    ctx.align_batch(mid, rd[:B], w.T_init[:B], raise_on_error=False)
    torch.cuda.synchronize(); print('call', k, 'ms', round((time.perf_counter()-t0)*1e3,2), file=sys.stderr)
    ctx.debug_counters()
