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
        wgt = np.ones_like(d2)
        rf = int(chain.get("robust_fct", 0))
        if rf:
            # RobustOutlierFilter (the chain's distance filter then): scale from the median absolute deviation of the finite squared
            # distances (elements at index size // 2), e2 = d2 / scale^2, the weight functions of the published source
            s2 = 1.0
            if int(chain.get("robust_scale", 1)) == 1:
                med = np.partition(vals, nf // 2)[nf // 2]
                dev = np.abs(vals - med)
                s2 = float(np.partition(dev, nf // 2)[nf // 2])
            kt = float(chain.get("robust_tuning", 1.0)); k2 = kt * kt
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                e2 = d2 / s2
                wgt = {1: lambda: 1.0 / (1.0 + e2 / k2), 2: lambda: np.exp(-e2 / k2), 3: lambda: np.where(e2 >= kt, 4.0 * k2 / (kt + e2) ** 2, 1.0),
                       4: lambda: k2 / (kt + e2) ** 2, 5: lambda: np.where(e2 >= k2, 0.0, (1.0 - e2 / k2) ** 2),
                       6: lambda: np.where(e2 >= k2, kt / np.sqrt(e2), 1.0), 7: lambda: 1.0 / np.sqrt(e2)}[rf]()
            ap = float(chain.get("robust_approx", 0.0))
            if ap > 0 and np.isfinite(ap):
                wgt = np.where(e2 >= ap * ap, 0.0, wgt)
            wgt = np.where(finite, wgt, 0.0)
            keep = finite & (wgt != 0)
            limit = np.inf
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


def np_icp_ex(reading, ref, nrm, T_init, chain, reading_nrm=None):
    """The same float64 restatement with the other modules the chain's slots may hold:
    KDTreeMatcher.knn > 1 (matches knn x N; the quantile runs over all knn * N distances, every pair is a constraint),
    PointToPointErrorMinimizer (weighted Kabsch through numpy.linalg.svd; residual = sum of |p - q|; zero covariance),
    SurfaceNormalOutlierFilter (reading normal, rotated with the reading, against the matched reference normal) and
    BoundTransformationChecker (angle and translation of the accumulated correction; exceeding either is an error),
    PointToPlaneErrorMinimizer{force4DOF} (the increment is a rotation about z and a translation: a 4 x 4 system)."""
    K = int(chain.get("knn", 1))
    p2point = int(chain.get("error_minimizer", 0)) in (1, 3)       # (3: PointToPointWithCov -- the same solve, the Censi covariance kept)
    p2p_cov = int(chain.get("error_minimizer", 0)) == 3
    force4dof = int(chain.get("error_minimizer", 0)) == 2          # PointToPlaneErrorMinimizer{force4DOF}: [rz tx ty tz] only
    max_angle = float(chain.get("normal_max_angle", 0.0))
    b_rot, b_tr = float(chain.get("bound_max_rot", 0.0)), float(chain.get("bound_max_trans", 0.0))
    rd = reading.astype(np.float64)
    rn = None if reading_nrm is None else reading_nrm.astype(np.float64)
    ref = ref.astype(np.float64)
    nrm = nrm.astype(np.float64)
    tree = cKDTree(ref)
    T = np.array(T_init, dtype=np.float64)
    T_iter = np.eye(4)
    quats = [np.array([1.0, 0, 0, 0])]
    trans = [np.zeros(3)]
    out = dict(converged=False, max_iter_reached=False, status=0)
    it = 0
    N = rd.shape[0]
    while True:
        Tc = T_iter @ T
        p = rd @ Tc[:3, :3].T + Tc[:3, 3]
        d, idx = tree.query(p, k=K)
        d = d.reshape(N, K); idx = idx.reshape(N, K)
        d2 = d * d
        finite = d2 <= chain["max_dist"] ** 2
        vals = d2[finite]
        nf = vals.size
        k = min(int(nf * chain["trim_ratio"]), nf - 1) if chain["trim_ratio"] < 1 else nf - 1
        limit = np.partition(vals, k)[k]
        keep = finite & (d2 <= limit)
        wgt = np.ones_like(d2)
        rf = int(chain.get("robust_fct", 0))
        if rf:
            # RobustOutlierFilter (the chain's distance filter then): scale from the median absolute deviation of the finite squared
            # distances (elements at index size // 2), e2 = d2 / scale^2, the weight functions of the published source
            s2 = 1.0
            if int(chain.get("robust_scale", 1)) == 1:
                med = np.partition(vals, nf // 2)[nf // 2]
                dev = np.abs(vals - med)
                s2 = float(np.partition(dev, nf // 2)[nf // 2])
            kt = float(chain.get("robust_tuning", 1.0)); k2 = kt * kt
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                e2 = d2 / s2
                wgt = {1: lambda: 1.0 / (1.0 + e2 / k2), 2: lambda: np.exp(-e2 / k2), 3: lambda: np.where(e2 >= kt, 4.0 * k2 / (kt + e2) ** 2, 1.0),
                       4: lambda: k2 / (kt + e2) ** 2, 5: lambda: np.where(e2 >= k2, 0.0, (1.0 - e2 / k2) ** 2),
                       6: lambda: np.where(e2 >= k2, kt / np.sqrt(e2), 1.0), 7: lambda: 1.0 / np.sqrt(e2)}[rf]()
            ap = float(chain.get("robust_approx", 0.0))
            if ap > 0 and np.isfinite(ap):
                wgt = np.where(e2 >= ap * ap, 0.0, wgt)
            wgt = np.where(finite, wgt, 0.0)
            keep = finite & (wgt != 0)
            limit = np.inf
        if rn is not None and max_angle > 0:
            a = rn @ Tc[:3, :3].T
            a = a / np.linalg.norm(a, axis=1, keepdims=True)
            bq = nrm[np.where(finite, idx, 0)]
            bq = bq / np.linalg.norm(bq, axis=2, keepdims=True)
            keep &= np.einsum("ni,nki->nk", a, bq) >= math.cos(max_angle)
        ii, kk = np.nonzero(keep)
        pk, qk, nk = p[ii], ref[idx[ii, kk]], nrm[idx[ii, kk]]
        wk = wgt[ii, kk]
        if p2point:
            mp, mq = pk.mean(0), qk.mean(0)
            M = (qk - mq).T @ (pk - mp)
            U, S, Vt = np.linalg.svd(M)
            R = U @ Vt
            if np.linalg.det(R) < 0:
                Vt = Vt.copy(); Vt[2] *= -1
                R = U @ Vt
            dT = np.eye(4); dT[:3, :3] = R; dT[:3, 3] = mq - R @ mp
            residual = float(np.sum(np.linalg.norm(pk - qk, axis=1)))
        else:
            e = np.sum(nk * (pk - qk), axis=1)
            J = np.column_stack([np.cross(pk, nk), nk])
            if rf:
                dT = rodrigues(np.linalg.solve(J.T @ (J * wk[:, None]), -J.T @ (wk * e)))
            elif force4dof:
                J4 = J[:, 2:]                                       # the z component of p x n, and n
                x4 = np.linalg.solve(J4.T @ J4, -J4.T @ e)
                dT = rodrigues(np.concatenate([[0.0, 0.0], x4]))
            else:
                dT = rodrigues(np.linalg.solve(J.T @ J, -J.T @ e))
            residual = float(np.sum(wk * e * e))
        T_iter = dT @ T_iter
        it += 1
        out.update(overlap=float(wk.sum()) / (N * K), residual=residual, trim_limit=float(limit), n_kept=int(keep.sum()), n_finite=int(nf))
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
        if not out["max_iter_reached"] and (b_rot > 0 or b_tr > 0):
            ang = abs(quat_angle(quats[-1], quats[0]))
            if (b_rot > 0 and ang > b_rot) or (b_tr > 0 and np.linalg.norm(trans[-1]) > b_tr):
                out["status"] = 7
                break
        if stop:
            mean = ref.mean(axis=0)
            out["cov"] = np.zeros((6, 6)) if (p2point and not p2p_cov) else censi_cov(pk - mean, qk - mean, nk, dT, chain["sensor_std_dev"])
            break
    out["T"] = T_iter @ T if out["status"] == 0 else np.eye(4)
    out["iterations"] = it
    return out


def main_variants():
    """tests/golden/chain_variants_small.npz: one scan-to-map problem through the chain with its other modules"""
    here = os.path.dirname(os.path.abspath(__file__))
    w = synth.make_scan_to_map(n_scan=3000, n_map=16000, n_queries=1, n_map_poses=4, rings=16)
    rd, T0 = w.scans_xyz[0], w.T_init[0]
    # reading normals for the SurfaceNormalOutlierFilter: the normals of the reading's nearest map points at the true pose,
    # expressed in the reading's frame, every seventh one turned by 60 degrees (pairs the filter must drop)
    Tt = w.T_truth[0]
    pt = rd.astype(np.float64) @ Tt[:3, :3].T + Tt[:3, 3]
    _, j = cKDTree(w.map_xyz.astype(np.float64)).query(pt)
    rn = w.map_nrm[j].astype(np.float64) @ Tt[:3, :3]
    c, s_ = math.cos(math.pi / 3), math.sin(math.pi / 3)
    Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]]); Rx = np.array([[1.0, 0, 0], [0, c, -s_], [0, s_, c]])
    rn[::7] = rn[::7] @ (Rz @ Rx).T
    rn = rn.astype(np.float32)
    fix = dict(map_xyz=w.map_xyz, map_nrm=w.map_nrm, reading=rd, reading_nrm=rn, T_init=T0, T_truth=Tt)
    variants = dict(knn3=dict(CHAIN, knn=3), p2point=dict(CHAIN, error_minimizer=1), p2point_knn2=dict(CHAIN, error_minimizer=1, knn=2),
                    normals=dict(CHAIN, normal_max_angle=0.5), bound_ok=dict(CHAIN, bound_max_rot=0.2, bound_max_trans=1.0),
                    bound_hit=dict(CHAIN, bound_max_rot=0.2, bound_max_trans=0.05), force4dof=dict(CHAIN, error_minimizer=2),
                    p2point_cov=dict(CHAIN, error_minimizer=3),
                    robust_cauchy=dict(CHAIN, trim_ratio=1.0, robust_fct=1, robust_tuning=1.0, robust_scale=1),
                    robust_huber=dict(CHAIN, trim_ratio=1.0, robust_fct=6, robust_tuning=2.0, robust_scale=1),
                    robust_tukey_none=dict(CHAIN, trim_ratio=1.0, robust_fct=5, robust_tuning=0.3, robust_scale=0, robust_approx=0.25))
    for name, ch in variants.items():
        r = np_icp_ex(rd, w.map_xyz, w.map_nrm, T0, ch, reading_nrm=rn if "normal_max_angle" in ch else None)
        for k in ("T", "iterations", "converged", "status", "overlap", "residual", "trim_limit", "n_kept", "n_finite"):
            fix[f"{name}_{k}"] = r[k]
        if "cov" in r:
            fix[f"{name}_cov"] = r["cov"]
        print(name, "status", r["status"], "iterations", r["iterations"], "kept", r["n_kept"], "overlap", round(r["overlap"], 4))
    np.savez_compressed(os.path.join(here, "chain_variants_small.npz"), **fix)


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


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "variants":
    main_variants()
    sys.exit(0)
if __name__ == "__main__":
    main()
