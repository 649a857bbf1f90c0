"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Bit-exact for indices / squared distances / trim
thresholds / transforms; final ICP transform within 1e-5 m and 1e-5 rad
(BASELINE.json north_star tolerance)."""
import math

import numpy as np
import pytest

from pgslam_amd import icp, synth

pytestmark = pytest.mark.gpu

TOL_TRANS = 1e-5   # metres
TOL_ROT = 1e-5     # radians

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)


def pose_error(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    tr = np.linalg.norm(d[:3, 3])
    c = min(1.0, max(-1.0, (np.trace(d[:3, :3]) - 1.0) / 2.0))
    return tr, math.acos(c)


@pytest.fixture(scope="module")
def small():
    return synth.make_scan_to_map(n_scan=6000, n_map=50_000, n_queries=3, n_map_poses=4, rings=16)


@pytest.fixture(scope="module")
def two_scans():
    return synth.make_two_scans(10_000, rings=16)


def rng_cloud(seed, n, scale=1.0):
    u = synth.uniform01(seed, 3 * n).reshape(n, 3)
    return ((u - 0.5) * scale).astype(np.float32)


# ---------------------------------------------------------------- matcher
@pytest.mark.parametrize("matcher", [icp.MATCHER_BRUTE, icp.MATCHER_GRID])
@pytest.mark.parametrize("max_dist", [2.0, 0.3, float("inf")])
def test_match_bit_exact_two_scans(ctx, oracle32, two_scans, matcher, max_dist):
    """BASELINE configs[0]: 10k vs 10k, ids and squared distances bit-exact."""
    w = two_scans
    ctx.set_params(**dict(CHAIN, max_dist=max_dist, matcher=matcher))
    mid = ctx.set_map(w["ref_xyz"], w["ref_nrm"], center=False)
    ids, d2 = ctx.match(mid, w["reading_xyz"], T=w["T_init"])
    q = oracle32.transform(w["T_init"], w["reading_xyz"])
    oid, od2 = oracle32.knn_brute(q, w["ref_xyz"], max_dist)
    ctx.destroy_map(mid)
    assert np.array_equal(ids, oid)
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))


@pytest.mark.parametrize("matcher", [icp.MATCHER_BRUTE, icp.MATCHER_GRID])
def test_match_ties_lowest_index_and_duplicates(ctx, oracle32, matcher):
    """Duplicate map points and exact ties across cell borders: lowest index wins."""
    base = rng_cloud(11, 500)
    m = np.concatenate([base, base[::-1], base[100:200]])              # many exact duplicates
    # lattice points: queries at cell-centre midpoints tie between neighbours
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    m = np.concatenate([m, (g * 0.125 - 0.5).astype(np.float32)])
    q = np.concatenate([base[:300], (g[:200] * 0.125 - 0.5 + 0.0625).astype(np.float32), rng_cloud(12, 500, 1.5)])
    ctx.set_params(**dict(CHAIN, max_dist=float("inf"), matcher=matcher))
    mid = ctx.set_map(m, None, center=False)
    ids, d2 = ctx.match(mid, q)
    oid, od2 = oracle32.knn_brute(q, m, np.inf)
    ctx.destroy_map(mid)
    assert np.array_equal(ids, oid)
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))


def test_match_max_dist_sentinels(ctx, oracle32):
    """Queries farther than maxDist from everything: id -1, dist +inf (Appendix B.5)."""
    m = rng_cloud(21, 2000)
    q = np.concatenate([rng_cloud(22, 100), rng_cloud(23, 100) + np.float32(10.0)])
    for matcher in (icp.MATCHER_BRUTE, icp.MATCHER_GRID):
        ctx.set_params(**dict(CHAIN, max_dist=0.5, matcher=matcher))
        mid = ctx.set_map(m, None, center=False)
        ids, d2 = ctx.match(mid, q)
        ctx.destroy_map(mid)
        oid, od2 = oracle32.knn_brute(q, m, 0.5)
        assert np.array_equal(ids, oid) and np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
        assert np.all(ids[100:] == -1) and np.all(np.isinf(d2[100:]))


def test_match_grid_equals_brute_on_scan_vs_map(ctx, oracle32, small):
    w = small
    q = w.scans_xyz[0]
    out = {}
    for matcher in (icp.MATCHER_BRUTE, icp.MATCHER_GRID):
        ctx.set_params(**dict(CHAIN, matcher=matcher))
        mid = ctx.set_map(w.map_xyz, w.map_nrm, center=False)
        out[matcher] = ctx.match(mid, q, T=w.T_init[0])
        ctx.destroy_map(mid)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    oid, od2 = oracle32.knn_kdtree(oracle32.transform(w.T_init[0], q), w.map_xyz, 2.0)
    assert np.array_equal(out[0][0], oid)
    assert np.array_equal(out[0][1].view(np.uint32), od2.view(np.uint32))


# ---------------------------------------------------------------- outlier filter
@pytest.mark.parametrize("ratio", [0.85, 0.5, 1.0, 0.999])
def test_trim_threshold_exact(ctx, oracle32, ratio):
    d2 = (synth.uniform01(31, 20_000) ** 2).astype(np.float32)
    d2[::97] = np.inf
    d2[5] = d2[6] = d2[7]                                        # ties
    ctx.set_params(**dict(CHAIN, trim_ratio=ratio))
    w, limit, nf = ctx.outlier_weights(d2)
    st, ow, olimit, onf = oracle32.trim_weights(d2, ratio)
    assert st == 0 and nf == onf
    assert np.float32(limit).view(np.uint32) == np.float32(olimit).view(np.uint32)
    assert np.array_equal(w, ow)


def test_trim_known_answer(ctx):
    """Appendix B.3: dists 0..9, ratio .85 -> index 8 -> limit 8 -> 9 kept; inf excluded from n."""
    ctx.set_params(**dict(CHAIN, trim_ratio=0.85))
    w, limit, nf = ctx.outlier_weights(np.arange(10, dtype=np.float32))
    assert limit == 8.0 and nf == 10 and w.sum() == 9
    d = np.concatenate([np.arange(10, dtype=np.float32), np.full(5, np.inf, np.float32)])
    w, limit, nf = ctx.outlier_weights(d)
    assert limit == 8.0 and nf == 10 and w.sum() == 9 and np.all(w[10:] == 0)
    with pytest.raises(icp.ConvergenceError):
        ctx.outlier_weights(np.full(8, np.inf, np.float32))


# ---------------------------------------------------------------- transform / map assembly
def test_transform_bit_exact(ctx, oracle32, small):
    T = synth.se3(0.3, -1.2, 0.05, 0.2, -0.03, 0.01)
    p = small.scans_xyz[1]
    assert np.array_equal(ctx.transform(T, p), oracle32.transform(T, p))
    n = small.scans_nrm[1]
    assert np.array_equal(ctx.transform(T, n, rotate_only=True), oracle32.transform(T, n, rotate_only=True))
    bad = T.copy()
    bad[0, 0] *= 1.5
    with pytest.raises(icp.PgicpError) as e:
        ctx.transform(bad, p)
    assert e.value.code == icp.ERR_NOT_RIGID


def test_build_local_map_bit_exact(ctx, oracle32, small):
    """LocalMap::BuildCloudFromData: reference cloud, then T_k * cloud_k appended (Appendix B.9)."""
    Ts = [np.eye(4), synth.se3(1.5, 0.1, 0, 0.02), synth.se3(-3.0, 0.2, 0, -0.03)]
    xs, ns = small.scans_xyz[:3], small.scans_nrm[:3]
    gx, gn = ctx.build_local_map(xs, ns, Ts)
    ox, on = oracle32.build_local_map(xs, ns, Ts)
    assert np.array_equal(gx, ox) and np.array_equal(gn, on)
    assert np.array_equal(gx[: xs[0].shape[0]], xs[0])


# ---------------------------------------------------------------- error elements
def test_error_stats_and_partial_chain(ctx, oracle32, small):
    w = small
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=False)
    T = w.T_init[1]
    moved = oracle32.transform(T, w.scans_xyz[1])
    ids, d2 = ctx.match(mid, moved)
    wts, limit, nf = ctx.outlier_weights(d2)
    ratio, resid, sys_ = ctx.error_stats(mid, moved, ids, wts)
    st, osys = oracle32.p2plane_system(moved, w.map_xyz, w.map_nrm, ids, wts)
    assert st == 0
    np.testing.assert_allclose(sys_, osys, rtol=1e-12, atol=1e-12)
    assert ratio == pytest.approx(osys[27] / moved.shape[0], rel=1e-14)
    # fused partial chain = Localizer::ComputeOverlapWith / LoopCloser::ComputeResidualError
    r2, e2 = ctx.partial_chain(mid, w.scans_xyz[1], T=T)
    o = oracle32.partial_chain(w.scans_xyz[1], w.map_xyz, w.map_nrm, T, **CHAIN)
    ctx.destroy_map(mid)
    assert o["status"] == 0
    assert r2 == pytest.approx(o["overlap"], rel=1e-14)
    assert e2 == pytest.approx(o["residual"], rel=1e-11)


# ---------------------------------------------------------------- full ICP
@pytest.mark.parametrize("matcher", [icp.MATCHER_GRID, icp.MATCHER_BRUTE])
def test_icp_scan_to_map_parity(ctx, oracle32, small, matcher):
    w = small
    ctx.set_params(**dict(CHAIN, matcher=matcher))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    for b in range(len(w.scans_xyz)):
        T, st = ctx.align(mid, w.scans_xyz[b], w.T_init[b])
        o = oracle32.icp(w.scans_xyz[b], w.map_xyz, w.map_nrm, w.T_init[b], **CHAIN)
        assert o["status"] == 0 and st["status"] == 0
        dt, dr = pose_error(o["T"], T)
        assert dt < TOL_TRANS and dr < TOL_ROT, (dt, dr)
        assert st["iterations"] == o["iterations"]
        assert st["converged"] == o["converged"] and st["max_iter_reached"] == o["max_iter_reached"]
        assert st["n_finite"] == o["n_finite"] and st["n_kept"] == o["n_kept"]
        assert st["overlap"] == pytest.approx(o["overlap"], rel=1e-12)
        assert st["residual"] == pytest.approx(o["residual"], rel=1e-6)
        assert np.float32(st["trim_limit"]) == np.float32(o["trim_limit"])
        np.testing.assert_allclose(st["cov"], o["cov"], rtol=1e-5, atol=1e-14)
        # and the answer is the right one
        gt, gr = pose_error(w.T_truth[b], T)
        assert gt < 0.02 and gr < 0.002
    ctx.destroy_map(mid)


def test_icp_batch_equals_single(ctx, small):
    w = small
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    singles = [ctx.align(mid, w.scans_xyz[b], w.T_init[b]) for b in range(3)]
    Tb, sb = ctx.align_batch(mid, w.scans_xyz, w.T_init)
    ctx.destroy_map(mid)
    for b in range(3):
        assert np.array_equal(Tb[b], singles[b][0])               # same kernels, same order: bit-identical
        assert sb[b]["iterations"] == singles[b][1]["iterations"]


def test_icp_pair_two_scans(ctx, oracle32, two_scans):
    """BASELINE configs[0] through ICP::operator()(reading, reference, T) (LoopCloser.hpp:98)."""
    w = two_scans
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    T, st = ctx.icp_pair(w["reading_xyz"], w["ref_xyz"], w["ref_nrm"], w["T_init"])
    o = oracle32.icp(w["reading_xyz"], w["ref_xyz"], w["ref_nrm"], w["T_init"], **CHAIN)
    dt, dr = pose_error(o["T"], T)
    assert dt < TOL_TRANS and dr < TOL_ROT
    assert st["iterations"] == o["iterations"]


def test_icp_fixed_iterations_counter_stop(ctx, oracle32, small):
    """Differential checker disabled -> the Counter stops the loop and raises the flag."""
    w = small
    prm = dict(CHAIN, max_iters=7, min_diff_rot=0.0, min_diff_trans=0.0)
    ctx.set_params(**dict(prm, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    T, st = ctx.align(mid, w.scans_xyz[0], w.T_init[0])
    ctx.destroy_map(mid)
    o = oracle32.icp(w.scans_xyz[0], w.map_xyz, w.map_nrm, w.T_init[0], **prm)
    assert st["iterations"] == 7 and st["max_iter_reached"] and not st["converged"]
    assert o["iterations"] == 7 and o["max_iter_reached"]
    dt, dr = pose_error(o["T"], T)
    assert dt < TOL_TRANS and dr < TOL_ROT


def test_icp_no_match_is_convergence_error(ctx, small):
    w = small
    ctx.set_params(**dict(CHAIN, max_dist=0.5, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    far = synth.se3(z=500.0)
    with pytest.raises(icp.ConvergenceError):
        ctx.align(mid, w.scans_xyz[0], far)
    ctx.destroy_map(mid)


def test_icp_double_precision(ctx, oracle64, small):
    w = small
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    mx, mn, rd = (a.astype(np.float64) for a in (w.map_xyz, w.map_nrm, w.scans_xyz[0]))
    mid = ctx.set_map(mx, mn, center=True)
    T, st = ctx.align(mid, rd, w.T_init[0])
    ids, d2 = ctx.match(mid, rd, T=w.T_init[0])
    ctx.destroy_map(mid)
    o = oracle64.icp(rd, mx, mn, w.T_init[0], **CHAIN)
    dt, dr = pose_error(o["T"], T)
    assert dt < TOL_TRANS and dr < TOL_ROT and st["iterations"] == o["iterations"]


# ---------------------------------------------------------------- golden fixtures + full size
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("b", [0, 1])
def test_gpu_matches_golden_fixture(ctx, b):
    """HIP path against the committed vectors of the independent numpy/scipy float64 ICP."""
    z = np.load(os.path.join(GOLD, "scan_to_map_small.npz"))
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(z["map_xyz"], z["map_nrm"], center=True)
    T, st = ctx.align(mid, z[f"reading{b}"], z[f"T_init{b}"])
    ids, d2 = ctx.match(mid, z[f"reading{b}"], T=z[f"T_init{b}"])
    ctx.destroy_map(mid)
    dt, dr = pose_error(z[f"T_final{b}"], T)
    assert dt < 1e-4 and dr < 1e-5          # float32 chain vs un-centred float64 chain (see tests/test_oracle.py)
    assert st["iterations"] == int(z[f"iterations{b}"]) and st["n_finite"] == int(z[f"n_finite{b}"])
    np.testing.assert_allclose(st["cov"], z[f"cov{b}"], rtol=1e-3, atol=1e-12)
    clear = (z[f"nn_gap{b}"] > 1e-3) & (z[f"nn_d{b}"] < 1.99)
    assert np.array_equal(ids[clear], z[f"nn_ids{b}"][clear])


@pytest.fixture(scope="module")
def full_size():
    """BASELINE.json configs[1] sizes: 100k-pt scans vs the 1M-pt map."""
    from bench import build_workload
    return build_workload(100_000, 1_000_000, 16)


def test_full_size_match_bit_exact(ctx, oracle32, full_size):
    w = full_size
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=False)
    for b, T in ((0, w.T_init[0]), (3, w.T_truth[3])):
        ids, d2 = ctx.match(mid, w.scans_xyz[b], T=T)
        oid, od2 = oracle32.knn_kdtree(oracle32.transform(T, w.scans_xyz[b]), w.map_xyz, 2.0)
        assert np.array_equal(ids, oid)
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    ctx.destroy_map(mid)


def test_full_size_icp_parity_and_properties(ctx, oracle32, full_size):
    w = full_size
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    om = oracle32.map_create(w.map_xyz, w.map_nrm)
    Ts, sts = ctx.align_batch(mid, w.scans_xyz[:4], w.T_init[:4])
    for b in range(4):
        # round 6: the pairs enter ONE reduction tree in ONE order on both sides (the device's sorting order, read back):
        # T_out, cov, residual, overlap of a 100 k x 1 M ICP are the oracle's doubles, bit for bit
        o = oracle32.icp_map(om, w.scans_xyz[b], w.T_init[b], pair_order=ctx.reading_order(len(w.scans_xyz[b]), problem=b), **CHAIN)
        dt, dr = pose_error(o["T"], Ts[b])
        assert dt < TOL_TRANS and dr < TOL_ROT, (b, dt, dr)
        assert sts[b]["iterations"] == o["iterations"] and sts[b]["converged"] == o["converged"]
        assert np.array_equal(Ts[b], o["T"]), (b, np.abs(Ts[b] - o["T"]).max())
        assert np.array_equal(np.asarray(sts[b]["cov"]).reshape(6, 6), o["cov"]) and sts[b]["residual"] == o["residual"] and sts[b]["overlap"] == o["overlap"]
    # ... and in scan order (sum_order = SCAN): a function of the inputs alone
    ctx.set_params(sum_order=icp.SUM_ORDER_SCAN)
    T_s, st_s = ctx.align(mid, w.scans_xyz[1], w.T_init[1])
    o = oracle32.icp_map(om, w.scans_xyz[1], w.T_init[1], **CHAIN)
    assert np.array_equal(T_s, o["T"]) and st_s["residual"] == o["residual"] and st_s["iterations"] == o["iterations"]
    ctx.set_params(sum_order=icp.SUM_ORDER_SORTED)
    oracle32.map_free(om)
    # size-independent properties: (1) a converged result is a fixed point up to the stop criterion:
    # restarting from it stops after smoothLength iterations and moves less than minDiffTransErr /
    # minDiffRotErr; (2) equivariance: moving the reading by S and the guess by
    # S^-1 describes the same problem up to float32 rounding of the moved reading
    T2, st2 = ctx.align(mid, w.scans_xyz[0], Ts[0])
    dt, dr = pose_error(Ts[0], T2)
    assert st2["iterations"] == 3 and dt < CHAIN["min_diff_trans"] and dr < CHAIN["min_diff_rot"]
    S = synth.se3(0.4, -0.3, 0.1, 0.05, 0.01, -0.02)
    moved = oracle32.transform(S, w.scans_xyz[1])
    T3, st3 = ctx.align(mid, moved, w.T_init[1] @ np.linalg.inv(S))
    dt, dr = pose_error(Ts[1], T3 @ S)
    assert dt < CHAIN["min_diff_trans"] and dr < CHAIN["min_diff_rot"]
    ctx.destroy_map(mid)


# ---------------------------------------------------------------- loop-closure batch (BASELINE configs[4], small)
def test_loop_closure_batch_against_oracle(ctx, oracle32):
    """Per pair: ICP::operator() (LoopCloser.hpp:98) + ComputeResidualError + CheckIcpResult, as one batch."""
    from pgslam_amd import loop_closure as lc
    ps = synth.make_pairs(5, n_pts=5000, n_keyframes=6, rings=16)
    cands = [lc.Candidate(from_id=10 + k, to_id=20 + k, reading=ps.reading_xyz[k], ref_xyz=ps.ref_xyz[k],
                          ref_nrm=ps.ref_nrm[k], T_init=ps.T_init[k]) for k in range(5)]
    cfg = lc.LoopClosureConfig(chain=dict(CHAIN, matcher=icp.MATCHER_GRID), residual_error_threshold=50.0)
    edges = lc.close_loops(ctx, cands, cfg)
    assert edges.shape == (5,) and np.all(edges["from_id"] == 10 + np.arange(5))
    for k in range(5):
        o = oracle32.icp(ps.reading_xyz[k], ps.ref_xyz[k], ps.ref_nrm[k], ps.T_init[k], **CHAIN)
        dt, dr = pose_error(o["T"], edges[k]["T_from_to"].reshape(4, 4))
        assert dt < TOL_TRANS and dr < TOL_ROT
        assert edges[k]["iterations"] == o["iterations"] and edges[k]["overlap"] == pytest.approx(o["overlap"], rel=1e-12)
        res = oracle32.partial_chain(ps.reading_xyz[k], ps.ref_xyz[k], ps.ref_nrm[k], o["T"], **CHAIN)
        assert edges[k]["residual"] == pytest.approx(res["residual"], rel=1e-3)
        expect = (not o["max_iter_reached"]) and o["overlap"] >= 0.8 and res["residual"] <= 50.0
        assert bool(edges[k]["accepted"]) == expect
        np.testing.assert_allclose(edges[k]["cov"].reshape(6, 6), o["cov"], rtol=1e-5, atol=1e-14)
    assert len(lc.accepted_constraints(edges)) == int(edges["accepted"].sum())


ROBUST = dict(trim_ratio=1.0, robust_fct=1, robust_tuning=2.0, robust_scale=1)
NOT_ROBUST = dict(robust_fct=0)


@pytest.mark.parametrize("extra", [NOT_ROBUST, ROBUST], ids=["trimmed", "robust"])
def test_fused_residual_pass_equals_the_separate_chain(ctx, oracle32, extra):
    """pgicp_align_residual_batch: the ICPs are those of pgicp_align_batch, bit for bit; the residual pass -- seeded with the
    last iteration's correspondences -- gives what the separate (unseeded) partial chain gives on the result.  Also with a
    RobustOutlierFilter in the chain (no threshold: every pair is resolved in the residual pass too)."""
    CHAIN = dict(globals()["CHAIN"], **extra)
    ps = synth.make_pairs(6, n_pts=5000, n_keyframes=6, rings=16)
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    rds = [ps.reading_xyz[k] for k in range(6)]
    T0 = [ps.T_init[k] for k in range(6)]
    T0[3] = T0[3] @ synth.se3(x=500.0)                     # a candidate that cannot be aligned: its ICP fails, its residual is +inf
    ids = ctx.set_maps([ps.ref_xyz[k] for k in range(6)], [ps.ref_nrm[k] for k in range(6)], center=True)
    Ta, sa = ctx.align_batch(ids, rds, T0, raise_on_error=False)
    Tb, sb, res, ratio, rst = ctx.align_residual_batch(ids, rds, T0)
    assert np.array_equal(Ta, Tb)
    for a, b in zip(sa, sb):
        assert (a["status"], a["iterations"], a["n_kept"], a["n_finite"], a["trim_limit"], a["overlap"], a["residual"]) == \
               (b["status"], b["iterations"], b["n_kept"], b["n_finite"], b["trim_limit"], b["overlap"], b["residual"])
        assert np.array_equal(a["cov"], b["cov"])
    ok = [k for k in range(6) if sa[k]["status"] == 0]
    assert 3 not in ok and len(ok) == 5 and np.isinf(res[3]) and rst[3] != 0
    r2, e2, s2 = ctx.partial_chain_batch([ids[k] for k in ok], [rds[k] for k in ok], [Ta[k] for k in ok])
    for j, k in enumerate(ok):
        assert rst[k] == 0 and s2[j] == 0
        # (the fused pass moves the PRE-TRANSFORMED reading by the iteration transform, the separate chain the reading by the
        # composed one: the last bits of a distance differ -- a count-based ratio does not see that, a mean robust weight does)
        rtol = 1e-9 if not extra["robust_fct"] else 1e-5
        assert res[k] == pytest.approx(e2[j], rel=1e-4) and ratio[k] == pytest.approx(r2[j], rel=rtol)
        po = oracle32.partial_chain(rds[k], ps.ref_xyz[k], ps.ref_nrm[k], Ta[k], **CHAIN)
        assert res[k] == pytest.approx(po["residual"], rel=1e-3) and ratio[k] == pytest.approx(po["overlap"], rel=rtol)
    for m in ids:
        ctx.destroy_map(m)
    ctx.set_params(trim_ratio=globals()["CHAIN"]["trim_ratio"], robust_fct=0) if extra["robust_fct"] else None


def test_batch_entry_points_equal_single_calls(ctx):
    """pgicp_map_create_batch / pgicp_partial_chain_batch give bit-identical results to the per-object calls."""
    ps = synth.make_pairs(4, n_pts=4000, n_keyframes=5, rings=16)
    ctx.set_params(**dict(CHAIN, matcher=icp.MATCHER_GRID))
    ids_b = ctx.set_maps([ps.ref_xyz[k] for k in range(4)], [ps.ref_nrm[k] for k in range(4)], center=True)
    ids_s = [ctx.set_map(ps.ref_xyz[k], ps.ref_nrm[k], center=True) for k in range(4)]
    assert len(set(ids_b + ids_s)) == 8
    Ts = [ps.T_true[k] if hasattr(ps, "T_true") else ps.T_init[k] for k in range(4)]
    rb, eb, sb = ctx.partial_chain_batch(ids_b, [ps.reading_xyz[k] for k in range(4)], Ts)
    assert np.all(sb == 0)
    for k in range(4):
        r1, e1 = ctx.partial_chain(ids_s[k], ps.reading_xyz[k], T=Ts[k])
        assert r1 == rb[k] and e1 == eb[k]
        ia, da = ctx.match(ids_b[k], ps.reading_xyz[k], T=Ts[k])
        i1, d1 = ctx.match(ids_s[k], ps.reading_xyz[k], T=Ts[k])
        np.testing.assert_array_equal(ia, i1)
        np.testing.assert_array_equal(da, d1)
    for m in ids_b + ids_s:
        ctx.destroy_map(m)


@pytest.mark.gpu
def test_upload_pipeline_matches_host_input():
    """pgicp_upload_*: readings transferred on the copy stream (pinned and pageable sources, sets reused round robin)
    give the bit-identical result of the same reading passed as a host buffer."""
    from pgslam_amd import icp, synth
    w = synth.make_scan_to_map(n_scan=5000, n_map=40000, n_queries=4, n_map_poses=3, rings=16)
    chain = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3,
                 sensor_std_dev=0.01)
    ctx = icp.Context(0, **chain)
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    ref = [ctx.align(mid, w.scans_xyz[q], w.T_init[q]) for q in range(4)]
    pinned = []
    for q in range(4):
        a = ctx.host_alloc(w.scans_xyz[q].shape, np.float32)
        a[...] = w.scans_xyz[q]
        pinned.append(a)
    # five uploads in a row (two sets, so three reuses), alternating pinned / pageable sources, one step ahead
    ups = [ctx.upload([pinned[q % 4]] if q % 2 == 0 else [w.scans_xyz[q % 4]], pinned=(q % 2 == 0)) for q in range(2)]
    for q in range(6):
        T, st = ctx.align(mid, ups[q % 2][0], w.T_init[q % 4])
        assert st["status"] == 0 and st["iterations"] == ref[q % 4][1]["iterations"]
        assert np.array_equal(T, ref[q % 4][0])
        nq = q + 2
        ups[q % 2] = ctx.upload([pinned[nq % 4]] if nq % 2 == 0 else [w.scans_xyz[nq % 4]], pinned=(nq % 2 == 0))
    # a whole batch in one upload, and an uploaded cloud as a reference cloud
    up = ctx.upload([w.scans_xyz[q] for q in range(4)])
    Tb, stb = ctx.align_batch(mid, up, [w.T_init[q] for q in range(4)])
    for q in range(4):
        assert np.array_equal(Tb[q], ref[q][0])
    for a in pinned:
        ctx.host_free(a)
    ctx.destroy_map(mid)
    ctx.close()


@pytest.mark.gpu
def test_rccl_allgather_edges_world_size_1():
    """pgicp_allgather_edges on the real RCCL path (communicator of one rank): the gathered list is the local one in
    candidate order, empty slots and unreported candidates are marked -1."""
    from pgslam_amd import icp, loop_closure as lc
    ctx = icp.Context(0)
    comm = icp.Comm(ctx, 1, 0, icp.comm_unique_id())
    rng = np.random.default_rng(5)
    n_total = 7
    mine = np.array([5, 0, 3], dtype=np.int32)
    local = np.zeros(3, dtype=lc.EDGE_DTYPE)
    local["from_id"] = [10, 11, 12]
    local["to_id"] = [20, 21, 22]
    local["T_from_to"] = rng.standard_normal((3, 16))
    local["cov"] = rng.standard_normal((3, 36))
    local["accepted"] = [1, 0, 1]
    out = comm.allgather_edges(local, mine, slots_per_rank=5, n_total=n_total)
    assert out.shape == (n_total,)
    for k, p in enumerate(mine):
        for f in ("from_id", "to_id", "accepted"):
            assert out[f][p] == local[f][k]
        assert np.array_equal(out["T_from_to"][p], local["T_from_to"][k]) and np.array_equal(out["cov"][p], local["cov"][k])
    rest = [i for i in range(n_total) if i not in mine]
    assert np.all(out["from_id"][rest] == -1) and np.all(out["status"][rest] == -1)
    # the block size every rank derives by itself
    assert icp.shard_slots([5, 1, 1, 1, 1, 1], 2) == 5 and icp.shard_slots([1] * 9, 4) == 3
    comm.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("trim_ratio, out_max", [(0.85, 0.12), (1.0, 0.2), (0.85, 5.0)])
def test_max_dist_outlier_filter_in_the_chain(ctx, oracle32, small, trim_ratio, out_max):
    """A MaxDistOutlierFilter next to (or instead of) the trimmed filter: its weights multiply in."""
    w = small
    chain = dict(CHAIN, trim_ratio=trim_ratio, outlier_max_dist=out_max)
    ctx.set_params(**dict(chain, matcher=icp.MATCHER_GRID))
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    for b in range(2):
        T, st = ctx.align(mid, w.scans_xyz[b], w.T_init[b])
        o = oracle32.icp(w.scans_xyz[b], w.map_xyz, w.map_nrm, w.T_init[b], **chain)
        assert o["status"] == 0 and st["status"] == 0
        dt, dr = pose_error(o["T"], T)
        assert dt < TOL_TRANS and dr < TOL_ROT, (dt, dr)
        assert st["iterations"] == o["iterations"] and st["n_kept"] == o["n_kept"] and st["n_finite"] == o["n_finite"]
        assert st["overlap"] == pytest.approx(o["overlap"], rel=1e-12)
        assert np.float32(st["trim_limit"]) == np.float32(o["trim_limit"])
    # ... and in the partial chain (ComputeOverlapWith / ComputeResidualError): raw index, as pgslam builds it there
    rid = ctx.set_map(w.map_xyz, w.map_nrm, center=False)
    ov, res = ctx.partial_chain(rid, w.scans_xyz[0], T=w.T_truth[0])
    po = oracle32.partial_chain(w.scans_xyz[0], w.map_xyz, w.map_nrm, w.T_truth[0], **chain)
    assert ov == pytest.approx(po["overlap"], rel=1e-12) and res == pytest.approx(po["residual"], rel=1e-6)
    ctx.destroy_map(rid)
    ctx.destroy_map(mid)
    ctx.set_params(**dict(CHAIN, outlier_max_dist=0.0))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [5, 1])
def test_host_may_look_at_the_iteration_flag_less_often(ctx, small, B):
    """check_every: the convergence checkers run on the device after every iteration; how often the HOST looks at their flag
    only decides how many (no-op) iterations it enqueues past the last one.  Transforms and statistics do not change.
    (B = 1: the solve kernel keeps the iteration stamp itself -- also in the no-op iterations past the converged one.)"""
    w = small
    rd = [w.scans_xyz[b % len(w.scans_xyz)] for b in range(B)]
    T0 = [w.T_init[b % len(w.scans_xyz)] @ synth.se3(x=0.01 * b, yaw=0.002 * b) for b in range(B)]
    m = ctx.set_map(w.map_xyz, w.map_nrm)
    ref = None
    for every in (1, 2, 3, 7):
        ctx.set_params(**CHAIN, check_every=every)
        Ts, st = ctx.align_batch(m, rd, T0, raise_on_error=False)
        key = (np.asarray(Ts).tobytes(), [(s["status"], s["iterations"], s["converged"], s["n_kept"], s["n_finite"], s["trim_limit"], s["overlap"],
                                           s["residual"]) for s in st])
        if ref is None:
            ref = key
        assert key == ref, every
    ctx.set_params(**CHAIN, check_every=1)
    ctx.destroy_map(m)


# ---------------------------------------------------------------- round 6: the overlap probe seeded from the ICP's correspondences
def test_seeded_partial_chain_equals_the_unseeded_one(oracle32):
    """pgicp_partial_chain_seeded (Localizer.hpp:210,231 -> 282-348: the overlap probe of a scan that has just been aligned): context A
    aligns a scan against the map of keyframes (k0, k1, k2); context B asks for the overlap with the candidate composition
    (k2, k1, k3) at the result.  The probe's matcher is seeded with A's correspondences moved through the shared keyframes' index
    offsets.  Seeds are candidates only: ratio and residual are the unseeded chain's doubles, the per-point matches the same --
    also with segments that lie (wrong offsets), a segment that is absent, and a reading A never aligned."""
    world = synth.make_world()
    poses = [synth.se3(x=-6.0 + 1.5 * k, yaw=np.deg2rad(1.5 * (k % 3 - 1))) for k in range(5)]
    kf = [synth.make_scan(world, poses[k], 12_000, 7100 + k, rings=16) for k in range(4)]
    ref_pose = poses[2]

    def assemble(order):
        xs, ns = [], []
        for k in order:
            x, n = synth.transform_cloud(synth.se3_inv(ref_pose) @ poses[k], kf[k][0], kf[k][1])
            xs.append(x); ns.append(n)
        return np.concatenate(xs).astype(np.float32), np.concatenate(ns).astype(np.float32), [len(x) for x in xs]
    order_a, order_b = [2, 1, 0], [2, 3, 1]
    xa, na, sizes_a = assemble(order_a)
    xb, nb, sizes_b = assemble(order_b)
    scan, _ = synth.make_scan(world, poses[4] @ synth.se3(x=-2.0), 10_000, 7200, rings=16)
    scan = scan.astype(np.float32)
    T0 = synth.se3_inv(ref_pose) @ poses[4] @ synth.se3(x=-2.0) @ synth.perturbation(41)
    A, B = icp.Context(0, **CHAIN), icp.Context(0, **CHAIN)
    ma = A.set_map(xa, na, center=True)
    mb = B.set_map(xb, nb, center=False)
    T, st = A.align(ma, scan, T0)
    assert st["status"] == 0
    plain = B.partial_chain(mb, scan, T=T)
    ids_plain, d2_plain = B.debug_last_matches(len(scan))
    start_a = np.concatenate([[0], np.cumsum(sizes_a)])
    start_b = np.concatenate([[0], np.cumsum(sizes_b)])
    dst = [int(start_b[order_b.index(k)]) if k in order_b else -1 for k in order_a]
    assert dst[2] == -1 and dst[0] == 0                       # k0 is not part of the candidate composition; k2 leads both
    seeded = B.partial_chain_seeded(mb, scan, T, A, start_a, dst)
    ids_s, d2_s = B.debug_last_matches(len(scan))

    def same(a, b):
        # the ratio (what the probe is asked for) is the same double; the residual is the same SUM taken in another order -- the seeded
        # call takes the reading in the ICP's sorting order (one set-up kernel instead of a sort), PGICP_SUM_ORDER_SORTED follows it
        return a[0] == b[0] and a[1] == pytest.approx(b[1], rel=1e-12)
    assert same(seeded, plain)
    kept = d2_plain <= np.quantile(d2_plain[np.isfinite(d2_plain)], 0.8)       # (inside the trim threshold: beyond it the capped search is lazy)
    assert np.array_equal(ids_s[kept & (ids_plain >= 0)], ids_plain[kept & (ids_plain >= 0)])
    # the oracle's chain on the same inputs
    o = oracle32.partial_chain(scan, xb, nb, T, **dict(CHAIN, center_reference=False))
    assert seeded[0] == pytest.approx(o["overlap"], rel=1e-12)
    # segments that lie: every seed points at some other point of the map -- candidates only
    lying = B.partial_chain_seeded(mb, scan, T, A, start_a, [5, 777, 31])
    assert same(lying, plain)
    # The seeded probe also starts with a search cap: 1.21 x the threshold the context's previous probe ended with (matcher mode 3:
    # the capped, lazily exact search of an ICP's later iterations).  A cap that is far too small -- the previous probe sat on the
    # map, this one is 40 cm off -- and one that is far too large cost time, never a result.
    T_off = T @ synth.se3(x=0.4, yaw=np.deg2rad(1.0))
    plain_off = B.partial_chain(mb, scan, T=T_off)
    assert plain_off[1] > 4.0 * plain[1]                      # (the residual says how far off: the threshold is several times the hint)
    B.partial_chain(mb, scan, T=T)                            # the hint: the small threshold
    assert same(B.partial_chain_seeded(mb, scan, T_off, A, start_a, dst), plain_off)
    ids_o, d2_o = B.debug_last_matches(len(scan))
    B.partial_chain(mb, scan, T=T_off)
    ids_po, d2_po = B.debug_last_matches(len(scan))
    kept_o = d2_po <= np.median(d2_po[np.isfinite(d2_po)])        # (well inside the trim threshold: exact in both calls)
    assert np.array_equal(ids_o[kept_o & (ids_po >= 0)], ids_po[kept_o & (ids_po >= 0)])
    assert same(B.partial_chain_seeded(mb, scan, T, A, start_a, dst), plain)       # the hint: the large threshold of T_off
    # a reading A did not align (another size): searched unseeded
    T2, _ = A.align(ma, scan[:5000], T0)
    other = B.partial_chain_seeded(mb, scan, T, A, start_a, dst)
    assert other == plain                                     # (unseeded: its own sort, the very same call)
    # and the seeds are worth something: fewer candidates are looked at (the fast matcher's launch is shorter) -- asserted as results
    # only; the timing is bench.py's (slam_100k leg)
    A.close(); B.close()
