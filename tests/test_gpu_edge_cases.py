"""Edge cases of the drop-in boundary on the GPU: ragged batches, tiny and degenerate clouds,
strided / homogeneous inputs, bad arguments.  Every result is checked against the oracle."""
import numpy as np
import pytest

from pgslam_amd import icp, synth

pytestmark = pytest.mark.gpu

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)


@pytest.fixture(scope="module")
def ctx():
    from pgslam_amd import icp
    c = icp.Context(0, **CHAIN)
    yield c
    c.close()


def pose_error(Ta, Tb):
    dT = np.linalg.inv(Ta) @ Tb
    return np.linalg.norm(dT[:3, 3]), np.arccos(np.clip((np.trace(dT[:3, :3]) - 1) / 2, -1, 1))


def test_ragged_batch_of_different_maps_and_sizes(ctx, oracle32):
    """One device batch mixing reading sizes (not multiples of 64) and maps of different density."""
    a = synth.make_two_scans(6000, rings=16)
    b = synth.make_two_scans(2500, rings=16)
    m_a = ctx.set_map(a["ref_xyz"], a["ref_nrm"])
    m_b = ctx.set_map(b["ref_xyz"][:1777], b["ref_nrm"][:1777])
    readings = [a["reading_xyz"][:5999], b["reading_xyz"][:1001], a["reading_xyz"][:63], b["reading_xyz"][:2500], a["reading_xyz"][:1]]
    maps = [m_a, m_b, m_a, m_b, m_a]
    refs = [(a["ref_xyz"], a["ref_nrm"]), (b["ref_xyz"][:1777], b["ref_nrm"][:1777])] * 3
    T0 = [a["T_init"], b["T_init"], a["T_init"], b["T_init"], a["T_init"]]
    Ts, st = ctx.align_batch(maps, readings, T0, raise_on_error=False)
    for k in range(5):
        o = oracle32.icp(readings[k], refs[k % 2][0], refs[k % 2][1], T0[k], **CHAIN)
        assert st[k]["status"] == o["status"], k
        if o["status"] == 0:
            dt, dr = pose_error(o["T"], Ts[k])
            assert dt < 1e-5 and dr < 1e-5, (k, dt, dr)
            assert st[k]["iterations"] == o["iterations"] and st[k]["n_finite"] == o["n_finite"]
        # each problem alone gives the same answer as inside the batch (the reading-sort bins, and with them
        # the order of the double-precision sums, are chosen per batch: equal to rounding, not bit for bit)
        T1, s1 = ctx.align_batch([maps[k]], [readings[k]], [T0[k]], raise_on_error=False)
        assert s1[0]["status"] == st[k]["status"]
        if st[k]["status"] == 0:
            np.testing.assert_allclose(T1[0], Ts[k], rtol=0, atol=1e-12)
            assert s1[0]["iterations"] == st[k]["iterations"]
    ctx.destroy_map(m_a); ctx.destroy_map(m_b)


def test_tiny_and_degenerate_maps(ctx, oracle32):
    one = np.array([[1.0, 2.0, 3.0]], dtype=np.float32)
    nrm = np.array([[0.0, 0.0, 1.0]], dtype=np.float32)
    m = ctx.set_map(one, nrm, center=False)
    q = np.array([[1.0, 2.0, 3.5], [50.0, 0.0, 0.0], [1.0, 2.0, 3.0]], dtype=np.float32)
    ids, d2 = ctx.match(m, q)
    oi, od = oracle32.knn_brute(q, one, 2.0)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(d2, od)
    ctx.destroy_map(m)
    # all points identical: every query ties on distance, the smallest index wins
    same = np.tile(one, (300, 1))
    m = ctx.set_map(same, np.tile(nrm, (300, 1)), center=False)
    ids, d2 = ctx.match(m, q)
    np.testing.assert_array_equal(ids, [0, -1, 0])
    ctx.destroy_map(m)
    # a line of points (flat bounding box in two axes)
    line = np.stack([np.linspace(-5, 5, 777), np.zeros(777), np.zeros(777)], 1).astype(np.float32)
    m = ctx.set_map(line, np.tile(nrm, (777, 1)), center=True)
    rng = np.random.default_rng(3)
    qq = (line[rng.integers(0, 777, 500)] + rng.normal(0, 0.3, (500, 3))).astype(np.float32)
    ids, d2 = ctx.match(m, qq)
    mean = oracle32.centroid(line)
    oi, od = oracle32.knn_brute((qq - mean).astype(np.float32), (line - mean).astype(np.float32), 2.0)
    np.testing.assert_array_equal(ids, oi)
    ctx.destroy_map(m)


def test_homogeneous_and_strided_inputs(ctx, oracle32):
    """libpointmatcher `features` are 4xN column-major = (N,4) rows with pad 1: stride 4, same result as (N,3)."""
    t = synth.make_two_scans(3000, rings=16)
    ref4 = np.concatenate([t["ref_xyz"], np.ones((3000, 1), np.float32)], 1)
    rd4 = np.concatenate([t["reading_xyz"], np.ones((3000, 1), np.float32)], 1)
    m3 = ctx.set_map(t["ref_xyz"], t["ref_nrm"])
    m4 = ctx.set_map(ref4, t["ref_nrm"])
    T3, s3 = ctx.align(m3, t["reading_xyz"], t["T_init"])
    T4, s4 = ctx.align(m4, rd4, t["T_init"])
    np.testing.assert_array_equal(T3, T4)
    assert s3["iterations"] == s4["iterations"]
    ctx.destroy_map(m3); ctx.destroy_map(m4)


def test_bad_arguments_are_reported_not_computed(ctx):
    from pgslam_amd import icp
    t = synth.make_two_scans(500, rings=16)
    bad = t["ref_xyz"].copy()
    bad[7, 1] = np.nan
    with pytest.raises(icp.PgicpError):
        ctx.set_map(bad, t["ref_nrm"])
    m = ctx.set_map(t["ref_xyz"], None)                       # match-only map: no normals
    with pytest.raises(icp.PgicpError):
        ctx.align(m, t["reading_xyz"], t["T_init"])            # point-to-plane needs them
    ids, _ = ctx.match(m, t["reading_xyz"])
    assert ids.shape == (500,)
    ctx.destroy_map(m)
    with pytest.raises(icp.PgicpError):
        ctx.align(12345, t["reading_xyz"], t["T_init"])        # unknown map id
    shear = np.eye(4); shear[0, 1] = 0.3
    with pytest.raises(icp.PgicpError):
        ctx.transform(shear, t["ref_xyz"])                     # RigidTransformation refuses a non-rigid matrix


def test_reading_with_points_far_outside_the_map(ctx, oracle32):
    """Queries far outside the grid's bounding box (clamped cells) and beyond maxDist keep the sentinels."""
    t = synth.make_two_scans(3000, rings=16)
    rd = t["reading_xyz"].copy()
    rd[::7] += np.float32(500.0)                               # far away: no neighbour within maxDist
    rd[3::11] *= np.float32(1.02)
    m = ctx.set_map(t["ref_xyz"], t["ref_nrm"])
    T, st = ctx.align(m, rd, t["T_init"])
    o = oracle32.icp(rd, t["ref_xyz"], t["ref_nrm"], t["T_init"], **CHAIN)
    dt, dr = pose_error(o["T"], T)
    assert dt < 1e-5 and dr < 1e-5 and st["n_finite"] == o["n_finite"] and st["iterations"] == o["iterations"]
    ctx.destroy_map(m)


def test_lazy_matcher_state_equals_oracle_every_iteration(ctx, oracle32):
    """Regression (sparse map, threshold that grows past the previous search cap): after every iteration
    count the kept pairs are the oracle's, id for id and bit for bit, and 'has a neighbour within maxDist'
    agrees for every point -- the quantities the lazily-exact matcher promises."""
    b = synth.make_two_scans(2500, rings=16)
    ref, nrm, rd, T0 = b["ref_xyz"][:1777], b["ref_nrm"][:1777], b["reading_xyz"], b["T_init"]
    m = ctx.set_map(ref, nrm)
    for it in (1, 2, 3, 4, 6):
        ctx.set_params(**dict(CHAIN, max_iters=it))
        T, st = ctx.align(m, rd, T0)
        gi, gd = ctx.debug_last_matches(rd.shape[0])
        o = oracle32.icp(rd, ref, nrm, T0, **dict(CHAIN, max_iters=it))
        assert st["iterations"] == o["iterations"] and st["n_finite"] == o["n_finite"] and st["n_kept"] == o["n_kept"]
        assert st["trim_limit"] == o["trim_limit"]
        np.testing.assert_array_equal(np.isfinite(gd), np.isfinite(o["last_d2"]))
        kept = o["last_d2"] <= o["trim_limit"]
        np.testing.assert_array_equal(gi[kept], o["last_ids"][kept])
        np.testing.assert_array_equal(gd[kept], o["last_d2"][kept])
        # whatever is not exact is an upper bound beyond the threshold
        loose = np.isfinite(gd) & ~kept
        assert np.all(gd[loose] >= o["last_d2"][loose]) and np.all(gd[loose] > o["trim_limit"])
    ctx.set_params(**CHAIN)
    ctx.destroy_map(m)


@pytest.mark.parametrize("n", [1, 2, 63, 2047, 2048, 2049, 4097, 30011])
def test_outlier_selection_sizes_and_ratios(ctx, oracle32, n):
    """The chip-wide trimmed-distance selection (histogram -> filter -> finish) at sizes around its 2048-point
    tiles and with several ratios, in one ragged batch: threshold, kept and finite counts equal the oracle's."""
    t = synth.make_two_scans(max(n, 3000), rings=16)
    ref, nrm, T0 = t["ref_xyz"], t["ref_nrm"], t["T_init"]
    rd = t["reading_xyz"][:n].copy()
    if n > 10:
        rd[::9] += np.float32(300.0)                          # some points without any neighbour
    m = ctx.set_map(ref, nrm)
    for ratio in (0.85, 0.5, 0.999, 1.0, 0.01):
        prm = dict(CHAIN, max_iters=2, trim_ratio=ratio)
        ctx.set_params(**prm)
        Ts, st = ctx.align_batch([m, m], [rd, t["reading_xyz"][:777]], [T0, T0], raise_on_error=False)
        o = oracle32.icp(rd, ref, nrm, T0, **prm)
        assert st[0]["status"] == o["status"], (n, ratio)
        if o["status"] == 0:
            assert st[0]["trim_limit"] == o["trim_limit"], (n, ratio)
            assert st[0]["n_kept"] == o["n_kept"] and st[0]["n_finite"] == o["n_finite"], (n, ratio)
    ctx.set_params(**CHAIN)
    ctx.destroy_map(m)


def test_outlier_selection_with_equal_distances(ctx, oracle32):
    """Every distance in ONE histogram bin (a plane of points matched from a parallel plane): the compact list
    of the selection is the whole reading."""
    g = np.arange(70, dtype=np.float32) * np.float32(0.125)
    xx, yy = np.meshgrid(g, g, indexing="ij")
    ref = np.stack([xx.ravel(), yy.ravel(), np.zeros(xx.size, dtype=np.float32)], 1).astype(np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (ref.shape[0], 1))
    rd = ref.copy()
    rd[:, 2] = np.float32(0.0625)                                # exactly representable: d2 = 2^-8 for every point
    m = ctx.set_map(ref, nrm, center=False)
    prm = dict(CHAIN, max_iters=1)
    ctx.set_params(**prm)
    T, st = ctx.align_batch([m], [rd], [np.eye(4)], raise_on_error=False)
    o = oracle32.icp(rd, ref, nrm, np.eye(4), **prm)
    assert st[0]["status"] == o["status"]
    assert st[0]["trim_limit"] == o["trim_limit"] == np.float32(0.0625) ** 2
    assert st[0]["n_kept"] == o["n_kept"] and st[0]["n_finite"] == o["n_finite"] == rd.shape[0]
    ctx.set_params(**CHAIN)
    ctx.destroy_map(m)


def test_outlier_selection_f64(ctx, oracle64):
    """Double precision: the selection finishes 52 remaining key bits on the compact list (five levels)."""
    t = synth.make_two_scans(5000, rings=16)
    ref, nrm, rd = t["ref_xyz"].astype(np.float64), t["ref_nrm"].astype(np.float64), t["reading_xyz"].astype(np.float64)
    m = ctx.set_map(ref, nrm)
    for ratio in (0.85, 0.3):
        prm = dict(CHAIN, max_iters=3, trim_ratio=ratio)
        ctx.set_params(**prm)
        T, st = ctx.align(m, rd, t["T_init"])
        o = oracle64.icp(rd, ref, nrm, t["T_init"], **prm)
        assert st["n_kept"] == o["n_kept"] and st["n_finite"] == o["n_finite"]
        np.testing.assert_allclose(st["trim_limit"], o["trim_limit"], rtol=1e-11)   # device vs host libm in the transforms
    ctx.set_params(**CHAIN)
    ctx.destroy_map(m)


def test_many_small_problems_in_one_batch(ctx, oracle32):
    """More than 128 problems: the matcher's queue has more than 1024 (problem, XCD) segments, so their offsets are
    scanned in several rounds.  Every problem must come out as it does alone."""
    t = synth.make_two_scans(3000, rings=16)
    m = ctx.set_map(t["ref_xyz"], t["ref_nrm"])
    P = 300
    rng = np.random.default_rng(11)
    readings, T0 = [], []
    for p in range(P):
        n = int(rng.integers(200, 900))
        start = int(rng.integers(0, 3000 - n))
        readings.append(np.ascontiguousarray(t["reading_xyz"][start:start + n]))
        T0.append(t["T_init"] @ synth.perturbation(p))
    Ts, st = ctx.align_batch([m] * P, readings, T0, raise_on_error=False)
    for p in list(range(0, P, 37)) + [P - 1]:
        o = oracle32.icp(readings[p], t["ref_xyz"], t["ref_nrm"], T0[p], **CHAIN)
        assert st[p]["status"] == o["status"], p
        if o["status"] == 0:
            dt, dr = pose_error(o["T"], Ts[p])
            assert dt < 1e-5 and dr < 1e-5, (p, dt, dr)
            assert st[p]["iterations"] == o["iterations"] and st[p]["n_finite"] == o["n_finite"] and st[p]["n_kept"] == o["n_kept"], p
    ctx.destroy_map(m)


def test_map_blocks_are_shared_between_contexts_and_huge_coordinates_are_refused():
    from pgslam_amd import icp
    """Advisor findings of round 1: (1) a map built by one context and released by another (pgicp_map_transfer, the
    streaming mapper's background rebuild) must give its block back to a pool the BUILDER draws from -- exercised here by
    build / transfer / destroy cycles whose results stay exact; (2) a cloud whose |coordinate| x point count overflows the
    centroid's fixed-point sum is refused, not silently mis-centred."""
    from pgslam_amd import synth
    w = synth.make_scan_to_map(n_scan=3000, n_map=20000, n_queries=1, n_map_poses=3, rings=16)
    chain = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3,
                 sensor_std_dev=0.01)
    serve, build = icp.Context(0, **chain), icp.Context(0, **chain)
    mid = serve.set_map(w.map_xyz, w.map_nrm, center=True)
    T_ref, st_ref = serve.align(mid, w.scans_xyz[0], w.T_init[0])
    serve.destroy_map(mid)
    for _ in range(6):
        b = build.set_map(w.map_xyz, w.map_nrm, center=True)
        m2 = serve.adopt_map(build, b)
        T, st = serve.align(m2, w.scans_xyz[0], w.T_init[0])
        assert np.array_equal(T, T_ref) and st["iterations"] == st_ref["iterations"]
        serve.destroy_map(m2)
    far = w.map_xyz.astype(np.float64) + np.array([4.2e7, 1.7e5, 4.8e6])       # 4.2e7 x 20 000 points > 5e11
    with pytest.raises(icp.PgicpError) as e:
        serve.set_map(far, w.map_nrm.astype(np.float64), center=True, dtype=np.float64)
    assert e.value.code == icp.ERR_ARG and "fixed-point" in str(e.value)
    serve.close()
    build.close()


def test_crowded_cells_overflow_the_item_pool(ctx, oracle32):
    """The fast matcher pools the neighbour rows' point ranges as items of 8 records (512 per wave); a wave whose ranges
    would overflow the pool takes larger items.  Here a map's bounding box is stretched by a few far points, so that the
    grid's cells are coarse and a dense cluster piles thousands of points into the rows around the queries: items of
    hundreds of records, several rounds.  Matches and the ICP must still be the oracle's, bit for bit / within 1e-5."""
    rng = np.random.default_rng(7)
    cluster = (rng.random((60_000, 3)) * np.array([1.2, 1.2, 0.4]) + np.array([10.0, -3.0, 0.5])).astype(np.float32)
    far = np.array([[-80, -80, -5], [80, 80, 5], [80, -80, 5], [-80, 80, -5]], dtype=np.float32)
    ref = np.concatenate([cluster, far])
    nrm = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (ref.shape[0], 1))
    q = (rng.random((5_000, 3)) * np.array([1.6, 1.6, 0.8]) + np.array([9.8, -3.2, 0.3])).astype(np.float32)
    m = ctx.set_map(ref, nrm, center=False)
    for T in (np.eye(4), synth.se3(x=0.07, y=-0.05, z=0.03, yaw=np.deg2rad(1.0))):
        ids, d2 = ctx.match(m, q, T=T)
        oid, od2 = oracle32.knn_kdtree(oracle32.transform(T, q), ref, 2.0)
        assert np.array_equal(ids, oid)
        assert np.array_equal(d2.view(np.uint32), od2.view(np.uint32))
    ctx.destroy_map(m)
    # and through the ICP loop (seeded passes, trimmed filter) on the centred map
    from test_gpu_matcher_state import check_state
    T0 = synth.se3(x=0.05, y=0.04, z=-0.02, yaw=np.deg2rad(0.8))
    check_state(ctx, oracle32, q, ref, nrm, T0, (1, 2, 4))


def test_debug_counters_on_a_context_that_has_not_aligned_anything():
    """a fresh context has no matcher counters yet: the call answers zeros instead of failing (bench.py's stream leg reads
    -- and thereby resets -- the selection's guess-miss counter before its first scan)"""
    c = icp.Context(0, max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
    out = c.debug_counters()
    assert list(out[:3]) == [0, 0, 0]
    c.close()


@pytest.mark.parametrize("offset", [0.0, 3.0e3, 4.0e5])
def test_double_clouds_far_from_the_origin_match_bit_for_bit(ctx, oracle64, offset):
    """PointMatcher<double> with the clouds in a frame whose origin is kilometres away (a UTM-like frame, the map NOT centred): the
    double fast matcher reads a float mirror of the map first (item_rounds_dp) -- at 400 km a float resolves 3 cm, so the
    prefilter must widen its margin with the coordinates' magnitude and still hand every possible winner to the double
    evaluation: ids and squared distances of every pair bit for bit the oracle's."""
    s = synth.make_two_scans(6000, rings=16)
    shift = np.array([offset, -0.7 * offset, 0.01 * offset])
    ref = s["ref_xyz"].astype(np.float64) + shift
    rd = s["reading_xyz"].astype(np.float64) + shift
    T0 = np.eye(4)
    T0[:3, 3] = [0.05, -0.03, 0.01]
    # T0 acts about the origin: a small rotation there would throw the cloud kilometres away, so translation only
    for max_dist in (2.0, 0.3):
        ctx.set_params(**dict(CHAIN, max_dist=max_dist))
        mid = ctx.set_map(ref, None, center=False, dtype=np.float64)
        ids, d2 = ctx.match(mid, rd, T=T0, dtype=np.float64)
        ctx.destroy_map(mid)
        q = oracle64.transform(T0, rd)
        oid, od2 = oracle64.knn_kdtree(q, ref, max_dist)
        assert np.array_equal(ids, oid)
        assert np.array_equal(d2.view(np.uint64), od2.view(np.uint64))
    ctx.set_params(**CHAIN)


def test_double_maps_of_one_batch_with_unbounded_max_dist(ctx, oracle64):
    """maxDist = inf, double clouds, two maps indexed in one batch (their points share one array): a query starts without any bound, so
    the float prefilter's first threshold is +inf -- it must still look only at its item's own records, not at the slots after them
    (the next map's points)."""
    a = synth.make_two_scans(3000, rings=16)
    b = synth.make_two_scans(1200, rings=8)
    refs = [a["ref_xyz"].astype(np.float64), b["ref_xyz"][:777].astype(np.float64) + np.array([0.3, -0.2, 0.1])]
    rds = [a["reading_xyz"].astype(np.float64), b["reading_xyz"].astype(np.float64)]
    ctx.set_params(**dict(CHAIN, max_dist=float("inf")))
    ids = ctx.set_maps(refs, None, center=True, dtype=np.float64)
    for k in range(2):
        got_ids, got_d2 = ctx.match(ids[k], rds[k], T=a["T_init"], dtype=np.float64)
        q = oracle64.transform(a["T_init"], rds[k])
        oid, od2 = oracle64.knn_kdtree(q, refs[k], np.inf)
        assert np.array_equal(got_ids, oid) and got_ids.max() < len(refs[k]) and got_ids.min() >= 0
        # (the map is centred: a distance is computed from centred coordinates, the last bits may differ from the oracle's uncentred ones)
        np.testing.assert_allclose(got_d2, od2, rtol=1e-9, atol=1e-18)
    for m in ids:
        ctx.destroy_map(m)
    ctx.set_params(**CHAIN)


@pytest.mark.parametrize("away", [5.0e3, 2.0e5])
def test_double_query_far_from_the_map_with_unbounded_max_dist(ctx, oracle64, away):
    """A double reading kilometres from its map (a bad initial pose) with maxDist = inf: the fast matcher's walk is float arithmetic
    (BK, k_match_common.inc), whose rounding at such offsets (2^-24 of 5 km is 0.3 mm per operation, of 200 km 12 mm) exceeds the grid's
    fixed 2 % margin: bk_margin widens every test by the query's own offset, so no row or window holding the true neighbour is
    pruned.  Part of the reading sits on the map (near queries), the rest far off: ids bit for bit the oracle's."""
    s = synth.make_two_scans(5000, rings=16)
    ref = s["ref_xyz"].astype(np.float64)
    rd = s["reading_xyz"].astype(np.float64).copy()
    rng = np.random.default_rng(12)
    far = rng.random(len(rd)) < 0.5
    rd[far] += np.array([away, 0.37 * away, -0.05 * away])
    ctx.set_params(**dict(CHAIN, max_dist=float("inf")))
    for center in (True, False):
        mid = ctx.set_map(ref, None, center=center, dtype=np.float64)
        ids, d2 = ctx.match(mid, rd, T=s["T_init"], dtype=np.float64)
        ctx.destroy_map(mid)
        q = oracle64.transform(s["T_init"], rd)
        oid, od2 = oracle64.knn_kdtree(q, ref, np.inf)
        assert np.array_equal(ids, oid), (center, int(np.sum(ids != oid)))
        np.testing.assert_allclose(d2, od2, rtol=1e-9)
    ctx.set_params(**CHAIN)
