"""Sensitivity envelope of the unpinned oracle: the variant builds of oracle/icp_oracle.c -- float
accumulation / float transforms as PointMatcher<float> does them (ORC_ACCUM_T), highest index wins ties (ORC_TIE_HIGH), fused
multiply-add in the rigid transform (ORC_FMA_TRANSFORM) -- against the default oracle on the three BASELINE configs whose
results must equal the reference's (/root/reference/src/pgslam/Localizer.hpp:126, LoopCloser.hpp:98).  Here at reduced sizes
(seconds); the full-size table is profiles/r05_sensitivity_envelope.json (tools/sensitivity_envelope.py), whose shape and
bounds are checked too.  The GPU tie census (the knn = 2 matcher at full size) is the -m gpu test at the end."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_variants_differ_from_the_oracle_only_where_they_should():
    """each flag changes exactly the arithmetic it names"""
    from oracle import Oracle
    rng = np.random.default_rng(3)
    o, of, oh, oa = Oracle(np.float32), Oracle(np.float32, "fma"), Oracle(np.float32, "tie_high"), Oracle(np.float32, "accum_t")
    pts = rng.normal(size=(2000, 3)).astype(np.float32) * 20
    T = np.eye(4)
    T[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    T[:3, 3] = [1.5, -2.25, 0.3]
    a, b = o.transform(T, pts), of.transform(T, pts)
    assert not np.array_equal(a, b) and np.max(np.abs(a - b)) < 1e-5           # FMA: last-bit differences only
    assert np.array_equal(o.transform(T, pts), oh.transform(T, pts))
    # ties: a map with every point duplicated -- lowest index by contract, highest index in the variant, same distances
    m = np.concatenate([pts[:500], pts[:500]])
    i0, d0 = o.knn_brute(pts[:300] + np.float32(0.01), m)
    i1, d1 = oh.knn_brute(pts[:300] + np.float32(0.01), m)
    assert np.array_equal(d0, d1) and np.all(i0 < 500) and np.array_equal(i1, i0 + 500)
    k0, _ = o.knn_kdtree(pts[:300] + np.float32(0.01), m)
    k1, _ = oh.knn_kdtree(pts[:300] + np.float32(0.01), m)
    assert np.array_equal(k0, i0) and np.array_equal(k1, i1)
    # float accumulation: the same system to float precision, not to double precision
    nrm = rng.normal(size=(500, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ids = np.arange(300, dtype=np.int32)
    w = np.ones(300, dtype=np.float32)
    _, s0 = o.p2plane_system(pts[:300] + np.float32(0.01), pts[:500], nrm, ids, w)
    _, s1 = oa.p2plane_system(pts[:300] + np.float32(0.01), pts[:500], nrm, ids, w)
    rel = np.abs(s0 - s1) / (np.abs(s0) + 1e-30)
    assert s0[28] == s1[28] == 300 and 0 < np.max(rel[:21]) < 1e-3


def test_envelope_at_reduced_size():
    import sensitivity_envelope as se
    rows = se.envelope(small=True)
    assert len(rows) == 3 * 4
    by = {(r["config"].split(" ")[0], r["variant"]): r for r in rows}
    for cfg in ("configs[0]", "configs[1]", "configs[4]"):
        # the tie rule is moot on these clouds, and a fused transform moves the result by less than the tolerance
        assert by[(cfg, "tie_high")]["max_dt_m"] < 1e-12 and by[(cfg, "tie_high")]["last_iteration_ids_differing"] == 0
        assert by[(cfg, "fma")]["within_1e5"], by[(cfg, "fma")]
        # float accumulation: same iteration count and verdicts; the distance is REPORTED (the live risk of DESIGN.md section 2)
        assert by[(cfg, "accum_t")]["max_d_iterations"] == 0 and by[(cfg, "accum_t")]["same_status_and_converged"]
        assert by[(cfg, "accum_t")]["max_dt_m"] < 1e-4 and by[(cfg, "accum_t")]["max_dr_rad"] < 1e-4


def test_committed_full_size_envelope():
    path = os.path.join(ROOT, "profiles", "r05_sensitivity_envelope.json")
    rec = json.load(open(path))
    assert not rec["small"]
    rows = rec["rows"]
    f32 = [r for r in rows if r["scalar"] == "f32"]
    assert {r["variant"] for r in f32} == {"accum_t", "tie_high", "fma", "all3"}
    assert len({r["config"] for r in f32}) == 3
    assert any("100000-pt scans vs 1000000-pt map" in r["config"] for r in f32)
    for r in f32:
        assert r["max_d_iterations"] == 0 and r["same_status_and_converged"]
        if r["variant"] in ("tie_high", "fma"):
            assert r["within_1e5"], r
        else:
            assert r["max_dt_m"] < 5e-5 and r["max_dr_rad"] < 1e-5, r       # beyond 1e-5 m at full size: named in DESIGN.md section 2


@pytest.mark.gpu
def test_tie_census_full_size():
    """the product's knn = 2 matcher counts exact ties on configs[1] and configs[4] at full size; wherever it finds one, the
    lower index comes first (the contract the brute-force oracle states)"""
    sys.path.insert(0, ROOT)
    import bench
    import tie_census as tc
    from pgslam_amd import icp, synth
    ctx = icp.Context(0, **tc.CHAIN)
    w = bench.build_workload(100_000, 1_000_000, 64)
    c1 = tc.census(ctx, w.map_xyz, w.map_nrm, w.scans_xyz[:2], w.T_init[:2])
    assert c1["queries"] == 200_000
    ps = synth.make_pairs(1, n_pts=100_000)
    c4 = tc.census(ctx, ps.ref_xyz[0], ps.ref_nrm[0], [ps.reading_xyz[0]], [ps.T_init[0]])
    assert c4["queries"] == 100_000
    # ordering of equal distances by the kernel, on a map that HAS ties (every point twice)
    m = np.concatenate([w.map_xyz[:50_000], w.map_xyz[:50_000]])
    ctx.set_params(knn=2)
    mid = ctx.set_map(m, None, center=False)
    ids, d2 = ctx.match(mid, w.map_xyz[:20_000] + np.float32(0.003))
    tie = d2[:, 0] == d2[:, 1]
    assert tie.sum() > 10_000 and np.all(ids[tie, 0] < ids[tie, 1])
    ctx.destroy_map(mid)
    ctx.close()
    print("tie census:", c1, c4)
