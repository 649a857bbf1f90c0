"""GPU parity of the chain's other modules (round 4): KDTreeMatcher.knn > 1, PointToPointErrorMinimizer,
SurfaceNormalOutlierFilter, BoundTransformationChecker -- the HIP path through the C ABI against the CPU oracle on the same
inputs, and against the independent float64 chain of tests/golden/make_golden.py (chain_variants_small.npz).
A pgslam user's YAML may name any of them (loadFromYaml at /root/reference/src/pgslam/Localizer.hpp:70, LoopCloser.hpp:73)."""
import math
import os

import numpy as np
import pytest

from pgslam_amd import icp, synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)
RESET = dict(knn=1, error_minimizer=0, bound_max_rot=0.0, bound_max_trans=0.0, normal_max_angle=0.0, outlier_max_dist=0.0,
             quantile_scale=1.0, robust_fct=0, robust_tuning=1.0, robust_scale=1, robust_approx=0.0)
VARIANTS = dict(knn3=dict(knn=3), p2point=dict(error_minimizer=1), p2point_knn2=dict(error_minimizer=1, knn=2),
                normals=dict(normal_max_angle=0.5), bound_ok=dict(bound_max_rot=0.2, bound_max_trans=1.0),
                bound_hit=dict(bound_max_rot=0.2, bound_max_trans=0.05), force4dof=dict(error_minimizer=2),
                p2point_cov=dict(error_minimizer=3),
                robust_cauchy=dict(trim_ratio=1.0, robust_fct=1, robust_tuning=1.0, robust_scale=1),
                robust_huber=dict(trim_ratio=1.0, robust_fct=6, robust_tuning=2.0, robust_scale=1),
                robust_tukey_none=dict(trim_ratio=1.0, robust_fct=5, robust_tuning=0.3, robust_scale=0, robust_approx=0.25))


def pose_error(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    c = min(1.0, max(-1.0, (np.trace(d[:3, :3]) - 1.0) / 2.0))
    return np.linalg.norm(d[:3, 3]), math.acos(c)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "chain_variants_small.npz"))


@pytest.mark.parametrize("k", [2, 3, 5, 8])
@pytest.mark.parametrize("max_dist", [2.0, 0.25, float("inf")])
def test_match_knn_bit_exact(ctx, oracle32, gold, k, max_dist):
    """knn neighbours per point in (distance, index) order, -1 / +inf where fewer lie within maxDist: bit for bit."""
    ctx.set_params(**{**CHAIN, **RESET, "max_dist": max_dist, "knn": k})
    mid = ctx.set_map(gold["map_xyz"], None, center=False)
    ids, d2 = ctx.match(mid, gold["reading"], T=gold["T_init"])
    ctx.destroy_map(mid)
    q = oracle32.transform(gold["T_init"], gold["reading"])
    oid, od2 = oracle32.knn_k(gold["map_xyz"], q, k, max_dist)
    assert ids.shape == (len(q), k)
    assert np.array_equal(ids, oid)
    assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    ctx.set_params(**dict(CHAIN, **RESET))


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_chain_variant_against_oracle_and_golden(ctx, oracle32, oracle64, gold, name, dtype):
    z = gold
    o = oracle32 if dtype == np.float32 else oracle64
    chain = dict(CHAIN, **VARIANTS[name])
    ctx.set_params(**dict(CHAIN, **RESET))
    ctx.set_params(**chain)
    rd, mx, mn = z["reading"].astype(dtype), z["map_xyz"].astype(dtype), z["map_nrm"].astype(dtype)
    rn = z["reading_nrm"].astype(dtype) if name == "normals" else None
    mid = ctx.set_map(mx, mn, center=True, dtype=dtype)
    r = o.icp(rd, mx, mn, z["T_init"], reading_nrm=rn, **chain)
    if name == "bound_hit":
        with pytest.raises(icp.ConvergenceError) as e:
            ctx.align(mid, rd, z["T_init"], dtype=dtype)
        assert e.value.code == icp.ERR_BOUND and r["status"] == 7 and int(z[f"{name}_status"]) == 7
    else:
        T, st = ctx.align(mid, rd, z["T_init"], dtype=dtype, normals=rn)
        assert st["status"] == 0 and r["status"] == 0
        dt, dr = pose_error(r["T"], T)
        assert dt < 1e-5 and dr < 1e-5, (dt, dr)
        assert st["iterations"] == r["iterations"] and st["converged"] == r["converged"]
        assert st["n_finite"] == r["n_finite"] and st["n_kept"] == r["n_kept"]
        if dtype == np.float32:
            assert np.float32(st["trim_limit"]) == np.float32(r["trim_limit"])
        else:       # (the solved transforms agree to ~1e-15: a double distance may differ in its last bits)
            assert st["trim_limit"] == pytest.approx(r["trim_limit"], rel=1e-9)
        assert st["overlap"] == pytest.approx(r["overlap"], rel=1e-12)
        assert st["residual"] == pytest.approx(r["residual"], rel=1e-6)
        if chain.get("error_minimizer", 0) == 1:
            assert not np.any(st["cov"])                      # PointToPoint: the base class's getCovariance
        else:                                                 # (PointToPointWithCov: the point-to-point result with the Censi estimate)
            if chain.get("error_minimizer", 0) == 3:
                assert np.any(st["cov"]) and np.allclose(st["cov"], st["cov"].T, rtol=1e-9, atol=1e-18)
            np.testing.assert_allclose(st["cov"], r["cov"], rtol=1e-5, atol=1e-14)
        # ... and the independent float64 chain
        gt, gr = pose_error(z[f"{name}_T"], T)
        assert gt < 1e-4 and gr < 1e-5, (gt, gr)
        assert st["iterations"] == int(z[f"{name}_iterations"]) and abs(st["n_kept"] - int(z[f"{name}_n_kept"])) <= 2
    ctx.destroy_map(mid)
    ctx.set_params(**dict(CHAIN, **RESET))


def test_knn_matcher_state_after_every_iteration(ctx, oracle32, gold):
    """knn = 3: ids and squared distances of ALL pairs bit for bit after 1, 2 and 3 iterations."""
    z = gold
    for iters in (1, 2, 3):
        chain = dict(CHAIN, knn=3, max_iters=iters, min_diff_rot=0.0, min_diff_trans=0.0)
        ctx.set_params(**dict(CHAIN, **RESET))
        ctx.set_params(**chain)
        mid = ctx.set_map(z["map_xyz"], z["map_nrm"], center=True)
        T, st = ctx.align(mid, z["reading"], z["T_init"])
        ids, d2 = ctx.debug_last_matches(len(z["reading"]))
        ctx.destroy_map(mid)
        r = oracle32.icp(z["reading"], z["map_xyz"], z["map_nrm"], z["T_init"], **chain)
        assert st["iterations"] == iters == r["iterations"]
        assert np.array_equal(ids, r["last_ids"])
        assert np.array_equal(d2.view(np.uint32), r["last_d2"].view(np.uint32))
        assert st["n_kept"] == r["n_kept"] and st["n_finite"] == r["n_finite"]
    ctx.set_params(**dict(CHAIN, **RESET))


def test_batch_of_variants_and_partial_chain(ctx, oracle32, gold):
    """knn = 2 in a batch (two problems, one map), and the partial chain (ComputeOverlapWith) with knn = 2."""
    z = gold
    chain = dict(CHAIN, knn=2)
    ctx.set_params(**dict(CHAIN, **RESET))
    ctx.set_params(**chain)
    mid = ctx.set_map(z["map_xyz"], z["map_nrm"], center=True)
    T0 = [z["T_init"], z["T_init"] @ synth.se3(x=0.03, yaw=0.004)]
    rds = [z["reading"], z["reading"][:2500]]
    Ts, sts = ctx.align_batch(mid, rds, T0)
    for b in range(2):
        r = oracle32.icp(rds[b], z["map_xyz"], z["map_nrm"], T0[b], **chain)
        dt, dr = pose_error(r["T"], Ts[b])
        assert dt < 1e-5 and dr < 1e-5
        assert sts[b]["iterations"] == r["iterations"] and sts[b]["n_kept"] == r["n_kept"]
    ctx.destroy_map(mid)
    rid = ctx.set_map(z["map_xyz"], z["map_nrm"], center=False)
    ov, res = ctx.partial_chain(rid, z["reading"], T=z["T_truth"])
    po = oracle32.partial_chain(z["reading"], z["map_xyz"], z["map_nrm"], z["T_truth"], **chain)
    assert ov == pytest.approx(po["overlap"], rel=1e-12) and res == pytest.approx(po["residual"], rel=1e-6)
    ctx.destroy_map(rid)
    ctx.set_params(**dict(CHAIN, **RESET))


def test_normals_filter_needs_reading_normals(ctx, gold):
    ctx.set_params(**{**CHAIN, **RESET, "normal_max_angle": 0.5})
    mid = ctx.set_map(gold["map_xyz"], gold["map_nrm"], center=True)
    with pytest.raises(icp.PgicpError):
        ctx.align(mid, gold["reading"], gold["T_init"])
    ctx.destroy_map(mid)
    ctx.set_params(**dict(CHAIN, **RESET))


def test_epsilon_above_zero_is_accepted_and_the_search_stays_exact(ctx, gold):
    """KDTreeMatcher.epsilon allows libnabo an approximate neighbour ((1 + epsilon) x the nearest distance); the exact one
    meets every allowance -- a configuration with epsilon 3.16 (libpointmatcher's example files) gives the results of epsilon 0"""
    z = gold
    ctx.set_params(**dict(CHAIN, **RESET))
    mid = ctx.set_map(z["map_xyz"], z["map_nrm"], center=True)
    T0, s0 = ctx.align(mid, z["reading"], z["T_init"])
    ids0, d0 = ctx.match(mid, z["reading"], z["T_init"])
    ctx.set_params(epsilon=3.16)
    T1, s1 = ctx.align(mid, z["reading"], z["T_init"])
    ids1, d1 = ctx.match(mid, z["reading"], z["T_init"])
    assert np.array_equal(T0, T1) and s0["iterations"] == s1["iterations"] and s0["n_kept"] == s1["n_kept"]
    assert np.array_equal(ids0, ids1) and np.array_equal(d0, d1)
    with pytest.raises(icp.PgicpError):
        ctx.set_params(epsilon=-0.5)
    ctx.set_params(epsilon=0.0)
    ctx.destroy_map(mid)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_robust_filter_stage_weights_bit_exact(ctx, oracle32, oracle64, dtype):
    """RobustOutlierFilter::compute at stage level (pgicp_outlier_weights under robust_* parameters): all seven functions, with
    the MAD scale and without, a cut-off, missing neighbours -- the weights bit for bit those of orc_robust_weights (welsch: to a few ulps, its exp() is a library function)."""
    o = oracle32 if dtype == np.float32 else oracle64
    rng = np.random.default_rng(77)
    d2 = (rng.gamma(1.5, 0.02, size=5001) ** 2).astype(dtype)
    d2[rng.integers(0, d2.size, 40)] = np.inf
    bits = np.uint32 if dtype == np.float32 else np.uint64
    for fct in range(1, 8):
        for scale in (1, 0):
            for approx in (0.0, 1.5):
                tuning = 1.0 if scale else 0.02
                ctx.set_params(**dict(CHAIN, **RESET))
                ctx.set_params(**dict(CHAIN, trim_ratio=1.0, robust_fct=fct, robust_tuning=tuning, robust_scale=scale, robust_approx=approx))
                w, limit, nf = ctx.outlier_weights(d2)
                ow, _ = o.robust_weights(d2, fct, tuning, scale, approx)
                assert nf == int(np.isfinite(d2).sum()) and np.isinf(limit)
                if fct == 2:    # welsch: exp() is not correctly rounded in either library (the device's and glibc's differ in the last bits)
                    np.testing.assert_allclose(w, ow, rtol=8 * np.finfo(dtype).eps, atol=np.finfo(dtype).tiny)
                else:
                    assert np.array_equal(w.view(bits), ow.view(bits)), (fct, scale, approx)
                assert np.all(w[np.isinf(d2)] == 0)
    with pytest.raises(icp.PgicpError):                         # a quantile filter beside it is refused
        ctx.set_params(**dict(CHAIN, trim_ratio=0.8, robust_fct=1))
    assert ctx.params.trim_ratio == 1.0 and ctx.params.robust_fct == 7          # (a refused setting leaves the mirror as it was)
    ctx.set_params(**dict(CHAIN, **RESET))
    with pytest.raises(icp.PgicpError):
        ctx.set_params(**dict(CHAIN, trim_ratio=1.0, robust_fct=1, knn=2))
    ctx.set_params(**dict(CHAIN, **RESET))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_robust_filter_in_a_batch_and_in_the_partial_chain(ctx, oracle32, oracle64, gold, dtype):
    """RobustOutlierFilter with several problems in flight (two readings of different sizes against one map: each its own two
    medians per iteration) and through the partial chain (ComputeOverlapWith / ComputeResidualError: the mean weight, the
    weighted residual)."""
    z = gold
    o = oracle32 if dtype == np.float32 else oracle64
    chain = dict(CHAIN, trim_ratio=1.0, robust_fct=1, robust_tuning=1.5, robust_scale=1)
    ctx.set_params(**dict(CHAIN, **RESET))
    ctx.set_params(**chain)
    mx, mn = z["map_xyz"].astype(dtype), z["map_nrm"].astype(dtype)
    mid = ctx.set_map(mx, mn, center=True, dtype=dtype)
    T0 = [z["T_init"], z["T_init"] @ synth.se3(x=0.03, yaw=0.004), z["T_init"] @ synth.se3(y=-0.02, yaw=-0.003)]
    rds = [z["reading"].astype(dtype), z["reading"][:2500].astype(dtype), z["reading"][700:].astype(dtype)]
    Ts, sts = ctx.align_batch(mid, rds, T0, dtype=dtype)
    for b in range(3):
        r = o.icp(rds[b], mx, mn, T0[b], **chain)
        dt, dr = pose_error(r["T"], Ts[b])
        assert dt < 1e-5 and dr < 1e-5, (b, dt, dr)
        assert sts[b]["iterations"] == r["iterations"] and sts[b]["n_finite"] == r["n_finite"] and sts[b]["n_kept"] == r["n_kept"]
        assert sts[b]["overlap"] == pytest.approx(r["overlap"], rel=1e-6 if dtype == np.float32 else 1e-12)
        assert 0.0 < sts[b]["overlap"] < 1.0 and np.isinf(sts[b]["trim_limit"])
    ctx.destroy_map(mid)
    rid = ctx.set_map(mx, mn, center=False, dtype=dtype)
    ov, res = ctx.partial_chain(rid, rds[0], T=z["T_truth"], dtype=dtype)
    po = o.partial_chain(rds[0], mx, mn, z["T_truth"], **chain)
    assert ov == pytest.approx(po["overlap"], rel=1e-6 if dtype == np.float32 else 1e-12) and res == pytest.approx(po["residual"], rel=1e-6)
    ctx.destroy_map(rid)
    ctx.set_params(**dict(CHAIN, **RESET))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_robust_filter_beside_the_other_filters(ctx, oracle32, oracle64, gold, dtype):
    """A MaxDistOutlierFilter and a SurfaceNormalOutlierFilter beside the robust one: the weights multiply -- pairs beyond maxDist
    and pairs whose normals disagree carry weight 0 whatever the M-estimator gives them; the medians are still over every
    finite distance."""
    z = gold
    o = oracle32 if dtype == np.float32 else oracle64
    mx, mn = z["map_xyz"].astype(dtype), z["map_nrm"].astype(dtype)
    rd, rn = z["reading"].astype(dtype), z["reading_nrm"].astype(dtype)
    for extra in (dict(outlier_max_dist=0.15), dict(normal_max_angle=0.5), dict(outlier_max_dist=0.3, normal_max_angle=0.8, error_minimizer=1)):
        chain = dict(CHAIN, trim_ratio=1.0, robust_fct=6, robust_tuning=1.5, robust_scale=1, **extra)
        ctx.set_params(**dict(CHAIN, **RESET))
        ctx.set_params(**chain)
        mid = ctx.set_map(mx, mn, center=True, dtype=dtype)
        use_n = "normal_max_angle" in extra
        T, st = ctx.align(mid, rd, z["T_init"], dtype=dtype, normals=rn if use_n else None)
        r = o.icp(rd, mx, mn, z["T_init"], reading_nrm=rn if use_n else None, **chain)
        ctx.destroy_map(mid)
        dt, dr = pose_error(r["T"], T)
        assert dt < 1e-5 and dr < 1e-5, (extra, dt, dr)
        assert st["iterations"] == r["iterations"] and st["n_finite"] == r["n_finite"] and st["n_kept"] == r["n_kept"], extra
        assert st["n_kept"] < st["n_finite"]                                 # the other filter did drop pairs
        assert np.float32(st["trim_limit"]) == np.float32(r["trim_limit"])
        assert st["overlap"] == pytest.approx(r["overlap"], rel=1e-6 if dtype == np.float32 else 1e-12)
    ctx.set_params(**dict(CHAIN, **RESET))
