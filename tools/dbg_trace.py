import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
from pgslam_amd import icp, synth
CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=2, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
b = synth.make_two_scans(2500, rings=16)
ref, nrm, rd, T0 = b["ref_xyz"][:1777], b["ref_nrm"][:1777], b["reading_xyz"][:2500], b["T_init"]
ctx = icp.Context(0, **CHAIN)
m = ctx.set_map(ref, nrm)
T, st = ctx.align(m, rd, T0)
gi, gd = ctx.debug_last_matches(2500)
print('final', gi[2360], gd[2360])
