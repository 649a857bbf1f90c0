import sys, numpy as np, torch, os
sys.path.insert(0,'.')
from bench import build_workload, CHAIN
from pgslam_amd import icp
w=build_workload(100000,1000000,16)
dev=torch.device('cuda',0)
rd=[torch.from_numpy(s).to(dev) for s in w.scans_xyz]
B=16
for iters in (1,2,3,5):
    ctx=icp.Context(0, **dict(CHAIN, max_iters=iters, min_diff_rot=0.0, min_diff_trans=0.0))
    mid=ctx.set_map(torch.from_numpy(w.map_xyz).to(dev), torch.from_numpy(w.map_nrm).to(dev))
    ctx.align_batch(mid, rd[:B], w.T_init[:B]); ctx.debug_counters()
    ctx.align_batch(mid, rd[:B], w.T_init[:B])
    print('cumulative over', iters, 'iterations, B=',B, file=sys.stderr); ctx.debug_counters()
    ctx.close()
