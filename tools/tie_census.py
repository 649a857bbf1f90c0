#!/usr/bin/env python3
"""Tie census (GPU; the product's knn = 2 matcher as the counter) -- how many reading points of the BASELINE workloads have TWO
nearest reference points at exactly the same squared distance?  libnabo's tie order is traversal dependent, the build's is
"lowest index" (DESIGN.md section 2): with zero exact ties the rule is moot.  Counted at the initial guess and at the converged
transform, in the centred frame the ICP matches in (ICP::operator() / setMap subtract the reference's mean), for
configs[1] (100 k-pt scans vs the 1 M-pt map), configs[2] (scans vs a 2 M-pt sliding map) and configs[4] (pairs of 100 k-pt clouds).

    python tools/tie_census.py [--out gpurun_out/tie_census.json] [--scans 16] [--pairs 16] [--no-stream]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)


def census(ctx, map_xyz, map_nrm, readings, T_inits):
    """ties of every reading against one map: dict(queries, ties_at_guess, ties_at_result, duplicate_map_points_hit)"""
    from pgslam_amd import icp
    ctx.set_params(**CHAIN, knn=1)
    mid = ctx.set_map(map_xyz, map_nrm, center=True)
    out = dict(queries=0, ties_at_guess=0, ties_at_result=0, ties_among_kept_at_result=0)
    for rd, Ti in zip(readings, T_inits):
        ctx.set_params(knn=1)
        T, st = ctx.align(mid, rd, Ti)
        ctx.set_params(knn=2)
        for key, Tm in (("ties_at_guess", Ti), ("ties_at_result", T)):
            ids, d2 = ctx.match(mid, rd, T=Tm)
            tie = (ids[:, 0] >= 0) & (ids[:, 1] >= 0) & (d2[:, 0] == d2[:, 1]) & (ids[:, 0] != ids[:, 1])
            out[key] += int(tie.sum())
            if key == "ties_at_result":
                out["ties_among_kept_at_result"] += int((tie & (d2[:, 0] <= np.float32(st["trim_limit"]))).sum())
        out["queries"] += int(rd.shape[0])
    ctx.set_params(knn=1)
    ctx.destroy_map(mid)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tie_census.json"))
    ap.add_argument("--scans", type=int, default=16)
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--no-stream", action="store_true")
    a = ap.parse_args()
    import bench
    from pgslam_amd import icp, synth
    ctx = icp.Context(0, **CHAIN)
    rec = {}
    w = bench.build_workload(100_000, 1_000_000, 64)
    rec["configs[1] 100k-pt scans vs 1M-pt map"] = census(ctx, w.map_xyz, w.map_nrm, w.scans_xyz[:a.scans], w.T_init[:a.scans])
    print(rec, flush=True)
    if not a.no_stream:
        # configs[2]: the first sliding map of the streaming leg (20 keyframes of 100 k points) and the scans that follow it
        n_prime, stride = 19, 3
        n_total = n_prime * stride + 9
        poses, odom, xyz, nrm = bench.build_drive(n_total, 100_000, 0.35)
        first = n_prime * stride
        kf = [first] + [k * stride for k in range(n_prime)]
        inv_ref = np.linalg.inv(poses[first])
        mx, mn = ctx.build_local_map([xyz[s] for s in kf], [nrm[s] for s in kf], [inv_ref @ poses[s] for s in kf])
        scans = [xyz[s] for s in range(first + 1, n_total)]
        guesses = [inv_ref @ poses[s] @ synth.perturbation(7000 + s) for s in range(first + 1, n_total)]
        rec["configs[2] 100k-pt scans vs 2M-pt sliding map"] = census(ctx, mx, mn, scans, guesses)
        print(rec, flush=True)
    kx, kn, kposes = bench.build_pairs(100_000)
    tot = dict(queries=0, ties_at_guess=0, ties_at_result=0, ties_among_kept_at_result=0)
    for p in range(a.pairs):
        i = p % len(kx)
        j = min(len(kx) - 1, i + 1 + (p // len(kx)) % 3) if i + 1 < len(kx) else i - 1
        T_true = synth.se3_inv(kposes[i]) @ kposes[j]
        c = census(ctx, kx[i], kn[i], [kx[j]], [T_true @ synth.perturbation(5000 + p)])
        for k in tot:
            tot[k] += c[k]
    rec["configs[4] pairs of 100k-pt clouds"] = tot
    print(rec, flush=True)
    out = dict(what="reading points whose two nearest reference points lie at exactly the same squared distance (different indices), counted with "
                    "the product's knn = 2 matcher in the centred frame; tools/tie_census.py", chain=CHAIN, census=rec)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
