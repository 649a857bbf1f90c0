#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the build container:
`python tests/golden/make_golden.py`).

The reference (Ellon/pgslam) holds no golden vectors and its arithmetic lives in
libpointmatcher, which is not installable here, so the
expected outputs are produced by an INDEPENDENT float64 restatement of the
chain written with numpy + scipy.spatial.cKDTree (SURVEY.md §8(c) "what pins the
build's results instead", item ii).  It shares no code with oracle/icp_oracle.c
or with the HIP path; tests compare both against these files.

Inputs come from the seeded synthetic generator (pgslam_amd/synth.py).
"""
import math
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pgslam_amd import synth  # noqa: E402

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)


def quat_from_R(R):
    w = math.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    x = math.copysign(math.sqrt(max(0.0, 1.0 + R[0, 0] - R[1, 1] - R[2, 2])) / 2.0, R[2, 1] - R[1, 2])
    y = math.copysign(math.sqrt(max(0.0, 1.0 - R[0, 0] + R[1, 1] - R[2, 2])) / 2.0, R[0, 2] - R[2, 0])
    z = math.copysign(math.sqrt(max(0.0, 1.0 - R[0, 0] - R[1, 1] + R[2, 2])) / 2.0, R[1, 0] - R[0, 1])
    q = np.array([w, x, y, z])
    return q / np.linalg.norm(q)


def quat_angle(a, b):
    # angular distance between two unit quaternions
    d = abs(float(np.dot(a, b)))
    v = math.sqrt(max(0.0, 1.0 - min(1.0, d) ** 2))
    return 2.0 * math.atan2(v, d)


def rodrigues(x):
    th = np.linalg.norm(x[:3])
    T = np.eye(4)
    if th > 0:
        k = x[:3] / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        T[:3, :3] = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)
    T[:3, 3] = x[3:]
    return T


def censi_cov(p, q, n, dT, sigma):
    beta = -math.asin(dT[2, 0])
    alpha = math.atan2(dT[2, 1], dT[2, 2])
    gamma = math.atan2(dT[1, 0] / math.cos(beta), dT[0, 0] / math.cos(beta))
    t = dT[:3, 3]
    rr = np.linalg.norm(p, axis=1)
    pd = p / rr[:, None]
    qr = np.linalg.norm(q, axis=1)
    qd = q / qr[:, None]
    na = n[:, 2] * pd[:, 1] - n[:, 1] * pd[:, 2]
    nb = n[:, 0] * pd[:, 2] - n[:, 2] * pd[:, 0]
    ng = n[:, 1] * pd[:, 0] - n[:, 0] * pd[:, 1]
    Rl = np.array([[1, -gamma, beta], [gamma, 1, -alpha], [-beta, alpha, 1]])
    E = np.sum(n * (p @ Rl.T + t - q), axis=1)
    Nr = np.sum(n * (pd @ Rl.T), axis=1)
    Nq = -np.sum(n * qd, axis=1)
    h = np.column_stack([n, rr * na, rr * nb, rr * ng])
    er = E + rr * Nr
    gr = np.column_stack([n * Nr[:, None], na * er, nb * er, ng * er])
    gq = np.column_stack([n * Nq[:, None], qr * na * Nq, qr * nb * Nq, qr * ng * Nq])
    H = h.T @ h
    G = gr.T @ gr + gq.T @ gq
    Hi = np.linalg.inv(H)
    return sigma ** 2 * Hi @ G @ Hi


def np_icp(reading, ref, nrm, T_init, chain):
    """float64 restatement of SURVEY.md Appendix A.2-A.9 (no centring: it is a
    mathematical no-op, it only changes float32 rounding)."""
    rd = reading.astype(np.float64)
    ref = ref.astype(np.float64)
    nrm = nrm.astype(np.float64)
    tree = cKDTree(ref)
    T = np.array(T_init, dtype=np.float64)
    T_iter = np.eye(4)
    quats = [np.array([1.0, 0, 0, 0])]
    trans = [np.zeros(3)]
    out = dict(converged=False, max_iter_reached=False)
    it = 0
    while True:
        Tc = T_iter @ T
        p = rd @ Tc[:3, :3].T + Tc[:3, 3]
        d, idx = tree.query(p)
        d2 = d * d
        finite = d2 <= chain["max_dist"] ** 2
        vals = d2[finite]
        nf = vals.size
        k = min(int(nf * chain["trim_ratio"]), nf - 1) if chain["trim_ratio"] < 1 else nf - 1
        limit = np.partition(vals, k)[k]
        keep = finite & (d2 <= limit)
        pk, qk, nk = p[keep], ref[idx[keep]], nrm[idx[keep]]
        e = np.sum(nk * (pk - qk), axis=1)
        J = np.column_stack([np.cross(pk, nk), nk])
        A = J.T @ J
        b = -J.T @ e
        x = np.linalg.solve(A, b)
        dT = rodrigues(x)
        T_prev_iter = T_iter
        T_iter = dT @ T_iter
        it += 1
        out.update(overlap=keep.sum() / rd.shape[0], residual=float(np.sum(e * e)), trim_limit=float(limit),
                   n_kept=int(keep.sum()), n_finite=int(nf))
        quats.append(quat_from_R(T_iter[:3, :3]))
        trans.append(T_iter[:3, 3].copy())
        stop = False
        if it >= chain["max_iters"]:
            out["max_iter_reached"] = True
            stop = True
        s = chain["smooth_length"]
        if len(quats) > s:
            r = np.mean([abs(quat_angle(quats[-1 - i], quats[-2 - i])) for i in range(s)])
            tt = np.mean([np.linalg.norm(trans[-1 - i] - trans[-2 - i]) for i in range(s)])
            if r < chain["min_diff_rot"] and tt < chain["min_diff_trans"]:
                out["converged"] = True
                stop = True
        if stop:
            # libpointmatcher's ICP runs in the frame of the mean-centred reference, so
            # the error elements its covariance estimator sees are expressed relative to the centroid
            mean = ref.mean(axis=0)
            out["cov"] = censi_cov(pk - mean, qk - mean, nk, dT, chain["sensor_std_dev"])
            out["first_ids"] = None
            break
    out["T"] = T_iter @ T
    out["iterations"] = it
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # ---- scan-to-map, small ------------------------------------------------
    w = synth.make_scan_to_map(n_scan=3000, n_map=16000, n_queries=2, n_map_poses=4, rings=16)
    fix = dict(map_xyz=w.map_xyz, map_nrm=w.map_nrm)
    for b in range(2):
        rd, T0 = w.scans_xyz[b], w.T_init[b]
        r = np_icp(rd, w.map_xyz, w.map_nrm, T0, CHAIN)
        # first-iteration correspondences (float64 kd-tree); ambiguity margin for float32 comparisons
        p = rd.astype(np.float64) @ T0[:3, :3].T + T0[:3, 3]
        d, idx = cKDTree(w.map_xyz.astype(np.float64)).query(p, k=2)
        fix[f"reading{b}"] = rd
        fix[f"T_init{b}"] = T0
        fix[f"T_truth{b}"] = w.T_truth[b]
        fix[f"T_final{b}"] = r["T"]
        fix[f"iterations{b}"] = r["iterations"]
        fix[f"converged{b}"] = r["converged"]
        fix[f"overlap{b}"] = r["overlap"]
        fix[f"residual{b}"] = r["residual"]
        fix[f"trim_limit{b}"] = r["trim_limit"]
        fix[f"n_kept{b}"] = r["n_kept"]
        fix[f"n_finite{b}"] = r["n_finite"]
        fix[f"cov{b}"] = r["cov"]
        fix[f"nn_ids{b}"] = idx[:, 0].astype(np.int32)
        fix[f"nn_d{b}"] = d[:, 0]
        fix[f"nn_gap{b}"] = d[:, 1] - d[:, 0]
    np.savez_compressed(os.path.join(here, "scan_to_map_small.npz"), **fix)
    # ---- configs[0]: two scans, 2k-pt version ------------------------------
    t = synth.make_two_scans(2000, rings=16)
    r = np_icp(t["reading_xyz"], t["ref_xyz"], t["ref_nrm"], t["T_init"], CHAIN)
    np.savez_compressed(os.path.join(here, "two_scans_small.npz"), reading=t["reading_xyz"], ref_xyz=t["ref_xyz"],
                        ref_nrm=t["ref_nrm"], T_init=t["T_init"], T_final=r["T"], iterations=r["iterations"],
                        overlap=r["overlap"], cov=r["cov"])
    # ---- surface normals: independent float64 restatement (scipy k-d tree + numpy eigh) -------
    # SurfaceNormalDataPointsFilter{knn=10, maxDist=2}: neighbours within maxDist, scatter about their
    # mean, eigenvector of the smallest eigenvalue.  `margin` is how far the 10th neighbour is from the
    # 11th (float32 evaluations may order a closer pair differently), `gap` the relative separation of
    # the two smallest eigenvalues (the normal is only defined where it is not tiny).
    s = synth.make_two_scans(3000, rings=16)
    xyz = s["ref_xyz"]
    knn, md = 10, 2.0
    x64 = xyz.astype(np.float64)
    d, idx = cKDTree(x64).query(x64, k=knn + 1)
    valid = d[:, :knn] <= md
    ids = np.where(valid, idx[:, :knn], -1).astype(np.int32)
    P = x64[np.where(valid, idx[:, :knn], 0)]
    cnt = valid.sum(1)
    mean = (P * valid[:, :, None]).sum(1) / cnt[:, None]
    D = (P - mean[:, None, :]) * valid[:, :, None]
    Cm = np.einsum("nki,nkj->nij", D, D)
    w, v = np.linalg.eigh(Cm)
    np.savez_compressed(os.path.join(here, "surface_normals_small.npz"), xyz=xyz, knn=knn, max_dist=md, ids=ids,
                        normals=v[:, :, 0], eigen_values=w, margin=d[:, knn] - d[:, knn - 1],
                        gap=(w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300), true_normals=s["ref_nrm"])
    print("wrote fixtures:", [f for f in os.listdir(here) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
