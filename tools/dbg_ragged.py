import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
from pgslam_amd import icp, synth
from oracle import Oracle
CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
o = Oracle(np.float32)
b = synth.make_two_scans(2500, rings=16)
ref, nrm, rd, T0 = b["ref_xyz"][:1777], b["ref_nrm"][:1777], b["reading_xyz"][:2500], b["T_init"]
ctx = icp.Context(0, **CHAIN)
m = ctx.set_map(ref, nrm)
for it in (1, 2, 3, 5, 8, 30):
    ctx.set_params(**dict(CHAIN, max_iters=it))
    T, st = ctx.align(m, rd, T0)
    r = o.icp(rd, ref, nrm, T0, **dict(CHAIN, max_iters=it))
    dT = np.linalg.inv(r["T"]) @ T
    print(it, 'iters', st["iterations"], r["iterations"], 'dt %.3e' % np.linalg.norm(dT[:3, 3]), 'n_finite', st["n_finite"], r["n_finite"],
          'kept', st["n_kept"], r["n_kept"], 'limit', st["trim_limit"], r["trim_limit"], 'resid', st["residual"], r["residual"])
for matcher in (icp.MATCHER_BRUTE,):
    ctx.set_params(**dict(CHAIN, matcher=matcher))
    T, st = ctx.align(m, rd, T0)
    r = o.icp(rd, ref, nrm, T0, **CHAIN)
    dT = np.linalg.inv(r["T"]) @ T
    print('brute: dt %.3e' % np.linalg.norm(dT[:3, 3]), st["iterations"], r["iterations"])
print('---- per-point comparison at max_iters=2')
ctx.set_params(**dict(CHAIN, max_iters=2, matcher=icp.MATCHER_GRID))
T, st = ctx.align(m, rd, T0)
gi, gd = ctx.debug_last_matches(rd.shape[0])
r = o.icp(rd, ref, nrm, T0, **dict(CHAIN, max_iters=2))
oi, od = r["last_ids"], r["last_d2"]
fin_g, fin_o = np.isfinite(gd), np.isfinite(od)
bad = np.nonzero(fin_g != fin_o)[0]
print('finite mismatch at', bad, 'gpu ids', gi[bad], 'gpu d2', gd[bad], 'oracle ids', oi[bad], 'oracle d2', od[bad], 'limit', r["trim_limit"])
kept = od <= r["trim_limit"]
print('kept pairs with different id:', np.sum(gi[kept] != oi[kept]), 'different d2:', np.sum(gd[kept] != od[kept]))
print('---- lazy unseeded path (partial chain) at the pose after iteration 1')
ctx.set_params(**dict(CHAIN, max_iters=1, matcher=icp.MATCHER_GRID))
T1, st = ctx.align(m, rd, T0)
ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
ctx.partial_chain(m, rd, T=T1)
gi, gd = ctx.debug_last_matches(rd.shape[0])
ei, ed = ctx.match(m, rd, T=T1)
print('partial chain state of 2360:', gi[2360], gd[2360], ' exact match:', ei[2360], ed[2360])
fin = np.isfinite(ed)
print('finite disagreement between lazy state and exact match:', np.nonzero(np.isfinite(gd) != fin)[0])
print('---- state after iteration 1 and 2')
for it in (1, 2):
    ctx.set_params(**dict(CHAIN, max_iters=it, matcher=icp.MATCHER_GRID))
    T, st = ctx.align(m, rd, T0)
    gi, gd = ctx.debug_last_matches(rd.shape[0])
    r = o.icp(rd, ref, nrm, T0, **dict(CHAIN, max_iters=it))
    print(it, 'gpu', gi[2360], gd[2360], 'oracle', r["last_ids"][2360], r["last_d2"][2360], 'limit', st["trim_limit"], 'counters', ctx.debug_counters())
