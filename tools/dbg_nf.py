import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
from pgslam_amd import icp
from oracle import Oracle
z=np.load('tests/golden/scan_to_map_small.npz'); b=1
CH=dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
o=Oracle(np.float32)
for it in range(1,6):
    ch=dict(CH, max_iters=it, min_diff_rot=0.0, min_diff_trans=0.0)
    ctx=icp.Context(0, **ch)
    mid=ctx.set_map(z['map_xyz'],z['map_nrm'])
    T,st=ctx.align(mid,z[f'reading{b}'],z[f'T_init{b}'])
    r=o.icp(z[f'reading{b}'],z['map_xyz'],z['map_nrm'],z[f'T_init{b}'],**ch)
    print(it,'gpu nf',st['n_finite'],'kept',st['n_kept'],'limit',st['trim_limit'],'| oracle nf',r['n_finite'],'kept',r['n_kept'],'limit',r['trim_limit'],'dbg',ctx.debug_counters())
    ctx.close()
