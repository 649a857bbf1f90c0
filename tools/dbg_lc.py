# loop-closure diagnostics: queue sizes of the lazy matcher per iteration (GPU box only)
import sys, numpy as np, torch
sys.path.insert(0, '.')
from pgslam_amd import icp, synth
from bench import CHAIN
ps = synth.make_pairs(8, n_pts=100000)
ctx = icp.Context(0, **CHAIN)
import os
ids = ctx.set_maps([ps.ref_xyz[k] for k in range(8)], [ps.ref_nrm[k] for k in range(8)])
for iters in (1, 2, 3, 5):
    ctx.set_params(**dict(CHAIN, max_iters=iters, min_diff_rot=0.0, min_diff_trans=0.0))
    T, st = ctx.align_batch(ids, [ps.reading_xyz[k] for k in range(8)], [ps.T_init[k] for k in range(8)])
    print(iters, 'last-iteration counters [queued, med survivors->slow2, ...]:', ctx.debug_counters(),
          'n_finite', [s['n_finite'] for s in st][:4], 'limit', [round(s['trim_limit'], 4) for s in st][:4],
          'overlap', [round(s['overlap'], 3) for s in st][:4])
