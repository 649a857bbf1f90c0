"""SurfaceNormalDataPointsFilter: the oracle against an independent
float64 restatement (golden fixture), the HIP kernel against the oracle."""
import os

import numpy as np
import pytest

from pgslam_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surface_normals_small.npz")


def align_sign(a, ref):
    s = np.sign(np.sum(a * ref, axis=1, keepdims=True))
    s[s == 0] = 1
    return a * s


def test_oracle_normals_match_golden(oracle32, oracle64):
    g = np.load(GOLD)
    knn, md = int(g["knn"]), float(g["max_dist"])
    for o, tol in ((oracle32, 2e-4), (oracle64, 1e-9)):
        r = o.surface_normals(g["xyz"], knn, md)
        safe = g["margin"] > 1e-5                      # neighbour set unambiguous at float32 resolution
        assert safe.mean() > 0.95
        np.testing.assert_array_equal(np.sort(r["ids"][safe], 1), np.sort(g["ids"][safe], 1))
        ok = safe & (g["gap"] > 1e-3)
        assert ok.mean() > 0.9
        dot = np.abs(np.sum(r["normals"][ok].astype(np.float64) * g["normals"][ok], axis=1))
        assert dot.min() > 1 - tol
        np.testing.assert_allclose(r["eigen_values"][ok], g["eigen_values"][ok], rtol=0, atol=tol * g["eigen_values"][ok, 2:3].max())
        # the estimate is a surface normal: it agrees with the analytic normal of the synthetic world
        # wherever the neighbourhood is planar (small smallest eigenvalue)
        planar = ok & (g["eigen_values"][:, 0] < 1e-3 * g["eigen_values"][:, 1])
        assert np.median(np.abs(np.sum(r["normals"][planar] * g["true_normals"][planar], axis=1))) > 0.99


def test_oracle_knn_k_is_lexicographic_and_bounded(oracle32):
    rng = np.random.default_rng(5)
    pts = np.round(rng.uniform(-1, 1, (600, 3)) * 8) / 8        # lattice: plenty of exact ties
    pts = pts.astype(np.float32)
    ids, d2 = oracle32.knn_k(pts, pts, 6, max_dist=0.3)
    for i in range(0, 600, 37):
        dx, dy, dz = (pts[i, 0] - pts[:, 0]), (pts[i, 1] - pts[:, 1]), (pts[i, 2] - pts[:, 2])
        dd = ((dx * dx + dy * dy) + dz * dz).astype(np.float32)
        order = np.lexsort((np.arange(600), dd))[:6]
        exp = np.where(dd[order] <= np.float32(0.3) ** 2, order, -1)
        np.testing.assert_array_equal(ids[i], exp)
        assert np.all(np.isinf(d2[i][exp < 0]))


def test_degenerate_neighbourhoods_take_the_library_defaults(oracle32):
    line = np.stack([np.linspace(0, 1, 50), np.zeros(50), np.zeros(50)], 1).astype(np.float32)   # rank 1 scatter
    r = oracle32.surface_normals(line, 5, 10.0)
    np.testing.assert_array_equal(r["normals"], np.tile([0, 1, 0], (50, 1)).astype(np.float32))
    lonely = np.array([[0, 0, 0], [100, 0, 0], [0, 100, 0]], dtype=np.float32)               # only itself within maxDist
    r = oracle32.surface_normals(lonely, 3, 1.0)
    np.testing.assert_array_equal(r["ids"], [[0, -1, -1], [1, -1, -1], [2, -1, -1]])
    np.testing.assert_array_equal(r["normals"], np.tile([0, 1, 0], (3, 1)).astype(np.float32))


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gctx():
    from pgslam_amd import icp
    c = icp.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("knn", [5, 10, 20])
def test_gpu_normals_match_oracle(gctx, oracle32, knn):
    s = synth.make_two_scans(10000, rings=16)
    xyz = s["ref_xyz"]
    o = oracle32.surface_normals(xyz, knn, 2.0)
    nrm, eig, ids, d2 = gctx.surface_normals(xyz, knn=knn, max_dist=2.0, want_eigen=True, want_ids=True)
    np.testing.assert_array_equal(ids, o["ids"])                  # bit-exact neighbours, (d2, index) order
    np.testing.assert_array_equal(d2, o["d2"])
    # same scatter sums in the same order and the same Jacobi sequence: identical up to the sign convention
    np.testing.assert_allclose(align_sign(nrm, o["normals"]), o["normals"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(eig, o["eigen_values"], rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
def test_gpu_normals_golden_and_edge_cases(gctx, oracle32):
    g = np.load(GOLD)
    nrm, ids, d2 = gctx.surface_normals(g["xyz"], knn=int(g["knn"]), max_dist=float(g["max_dist"]), want_ids=True)
    ok = (g["margin"] > 1e-5) & (g["gap"] > 1e-3)
    assert np.abs(np.sum(nrm[ok].astype(np.float64) * g["normals"][ok], axis=1)).min() > 1 - 2e-4
    # ties on a lattice, degenerate line, isolated points, unbounded maxDist
    rng = np.random.default_rng(5)
    pts = (np.round(rng.uniform(-1, 1, (600, 3)) * 8) / 8).astype(np.float32)
    o = oracle32.surface_normals(pts, 6, 0.3)
    n2, i2, dd2 = gctx.surface_normals(pts, knn=6, max_dist=0.3, want_ids=True)
    np.testing.assert_array_equal(i2, o["ids"])
    np.testing.assert_allclose(align_sign(n2, o["normals"]), o["normals"], atol=1e-6)
    line = np.stack([np.linspace(0, 1, 50), np.zeros(50), np.zeros(50)], 1).astype(np.float32)
    np.testing.assert_array_equal(gctx.surface_normals(line, knn=5, max_dist=10.0), np.tile([0, 1, 0], (50, 1)).astype(np.float32))
    lonely = np.array([[0, 0, 0], [100, 0, 0], [0, 100, 0]], dtype=np.float32)
    n3, i3, _ = gctx.surface_normals(lonely, knn=3, max_dist=1.0, want_ids=True)
    np.testing.assert_array_equal(i3, [[0, -1, -1], [1, -1, -1], [2, -1, -1]])
    o_inf = oracle32.surface_normals(pts, 8, np.inf)
    _, i4, _ = gctx.surface_normals(pts, knn=8, want_ids=True)
    np.testing.assert_array_equal(i4, o_inf["ids"])


@pytest.mark.gpu
def test_gpu_normals_feed_point_to_plane_icp(gctx, oracle32):
    """The purpose of the filter: a reference cloud without normals becomes usable for point-to-plane ICP."""
    from test_gpu_parity import CHAIN, pose_error
    t = synth.make_two_scans(10000, rings=16)
    est = gctx.surface_normals(t["ref_xyz"], knn=10, max_dist=2.0)
    gctx.set_params(**CHAIN)
    T_est, st = gctx.icp_pair(t["reading_xyz"], t["ref_xyz"], est, t["T_init"])
    dt, dr = pose_error(t["T_truth"], T_est)
    assert st["converged"] and dt < 0.05 and dr < 0.01
