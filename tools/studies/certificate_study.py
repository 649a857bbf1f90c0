"""Study (CPU, scipy): how often could a seeded pass of the matcher skip its search?

A query's previous search proved: every map point other than its match lies at >= r_other from the previous position.
If the query moved by delta since, and its old match now lies at d_new < r_other - delta, the old match is still THE
nearest neighbour and no search is needed.  r_other is at best the previous SECOND-nearest distance (a search that prunes
with the best distance proves less: rows are dropped as soon as their slab distance exceeds the best).
Prints, per iteration of the benchmark's ICP (100 k-pt scan, 1 M-pt map), the fraction of queries that qualify for
r_other = d2_prev and for r_other = min(d2_prev, d1_prev + g) with g = 0.5, 1, 2, 4 cm."""
import sys, os
sys.path.insert(0, '.')
import numpy as np
from scipy.spatial import cKDTree
from bench import build_workload, CHAIN
from pgslam_amd import synth
from oracle.oracle import Oracle

nq = 16
w = build_workload(100000, 1000000, nq)
orc = Oracle(np.float32)
tree = cKDTree(w.map_xyz.astype(np.float64))
for b in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    T0 = w.T_truth[b] @ synth.perturbation(b)
    r = orc.icp(w.scans_xyz[b], w.map_xyz, w.map_nrm, T0, trace=True, **CHAIN)
    Ts = [T0] + list(r["trace"])
    print(f"scan {b}: {r['iterations']} iterations, trim limit {np.sqrt(r['trim_limit']):.3f} m")
    p = w.scans_xyz[b].astype(np.float64)
    prev = None
    for k, T in enumerate(Ts[:-1]):
        q = p @ T[:3, :3].T + T[:3, 3]
        d, idx = tree.query(q, k=2, workers=8)
        line = f"  pass {k + 1}: median d1 {np.median(d[:, 0]) * 100:.2f} cm, median d2-d1 {np.median(d[:, 1] - d[:, 0]) * 100:.2f} cm"
        if prev is not None:
            qp, dp, ip = prev
            delta = np.linalg.norm(q - qp, axis=1)
            d_new = np.linalg.norm(q - tree.data[ip[:, 0]], axis=1)
            same = (idx[:, 0] == ip[:, 0]).mean()
            line += f", median move {np.median(delta) * 100:.2f} cm, same match {same * 100:.1f} %, certificate:"
            ok = d_new < dp[:, 1] - delta
            line += f" ideal {ok.mean() * 100:.1f} %"
            for g in (0.005, 0.01, 0.02, 0.04):
                ok = d_new < np.minimum(dp[:, 1], dp[:, 0] + g) - delta
                line += f", +{g * 100:.1f} cm {ok.mean() * 100:.1f} %"
        print(line)
        prev = (q, d, idx)
