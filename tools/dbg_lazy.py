import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from bench import build_workload, CHAIN
from pgslam_amd import icp
w=build_workload(100000,1000000,16)
dev=torch.device('cuda',0)
for iters in (1,2,3):
    ctx=icp.Context(0, **dict(CHAIN, max_iters=iters, min_diff_rot=0.0, min_diff_trans=0.0))
    mid=ctx.set_map(torch.from_numpy(w.map_xyz).to(dev), torch.from_numpy(w.map_nrm).to(dev))
    rd=[torch.from_numpy(s).to(dev) for s in w.scans_xyz]
    for B in (1,16):
        ctx.align_batch(mid, rd[:B], w.T_init[:B])
        torch.cuda.synchronize(); t=time.perf_counter()
        T,st=ctx.align_batch(mid, rd[:B], w.T_init[:B])
        dt=time.perf_counter()-t
        print('iters',iters,'B',B,'ms',dt*1e3,'dbg',ctx.debug_counters(),'nf',st[0]['n_finite'],'limit',st[0]['trim_limit'],'kept',st[0]['n_kept'])
    ctx.close()
