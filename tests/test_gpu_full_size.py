"""Parity at BASELINE.json's full sizes for the two configs that the other files only cover at toy sizes:

configs[2]  one 100k-pt scan late in the drive against a 20-keyframe / 2M-pt sliding local map: per-point matcher state
            after 1 and 2 iterations, bit for bit against the oracle (what tests/test_gpu_matcher_state.py does for
            configs[1]).  Reference calls: Localizer.hpp:126 (ICP against the local map), LocalMap.hpp:209-224 (the map).
configs[4]  100k-pt keyframe vs 100k-pt candidate map loop-closure pairs: ICP::operator() + ComputeResidualError +
            CheckIcpResult through the batch dispatcher, against the oracle (LoopCloser.hpp:83-110, 308-365).

The clouds are bench.py's own (same cache files), so the driver's bench run re-uses what these tests generated."""
import numpy as np
import pytest

from pgslam_amd import icp, synth

pytestmark = pytest.mark.gpu

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)
TOL_TRANS, TOL_ROT = 1e-5, 1e-5


def pose_error(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    return float(np.linalg.norm(d[:3, 3])), float(np.linalg.norm([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) / 2.0)


@pytest.fixture(scope="module")
def ctx():
    c = icp.Context(0, **CHAIN)
    yield c
    c.close()


def test_sliding_map_full_size_state(ctx, oracle32):
    from bench import build_drive
    from test_gpu_matcher_state import check_state
    capacity, stride, timed = 20, 3, 41
    n_total = (capacity - 1) * stride + timed
    poses, odom, xyz, nrm = build_drive(n_total, 100_000, 0.35)
    ref_s = n_total - 7                                        # the reference keyframe: late in the drive
    kf = [ref_s] + [ref_s - stride * k for k in range(1, capacity)]
    inv_ref = np.linalg.inv(poses[ref_s])
    mx, mn = oracle32.build_local_map([xyz[s] for s in kf], [nrm[s] for s in kf], [inv_ref @ poses[s] for s in kf])
    assert mx.shape[0] == capacity * 100_000
    # the device assembles the same map from the keyframe clouds (a12), bit for bit
    gx, gn = ctx.build_local_map([xyz[s] for s in kf], [nrm[s] for s in kf], [inv_ref @ poses[s] for s in kf])
    assert np.array_equal(gx.view(np.uint32), mx.view(np.uint32)) and np.array_equal(gn.view(np.uint32), mn.view(np.uint32))
    s = n_total - 1                                            # 2.1 m past the reference keyframe: part of it ahead of the map
    T0 = inv_ref @ poses[s] @ synth.se3(x=0.04, y=-0.03, yaw=np.deg2rad(0.3))
    check_state(ctx, oracle32, xyz[s], mx, mn, T0, (1, 2))


def test_loop_closure_pairs_full_size(ctx, oracle32):
    from bench import build_pairs
    from pgslam_amd import loop_closure as lc
    xyz, nrm, poses = build_pairs(100_000)
    pairs = [(3, 5, 5003), (10, 11, 5010)]
    cands = []
    for i, j, seed in pairs:
        T_true = synth.se3_inv(poses[i]) @ poses[j]
        cands.append(lc.Candidate(from_id=i, to_id=j, reading=xyz[j], ref_xyz=xyz[i], ref_nrm=nrm[i], T_init=T_true @ synth.perturbation(seed)))
    cfg = lc.LoopClosureConfig(chain=dict(CHAIN))
    edges = lc.close_loops(ctx, cands, cfg)
    for k, c in enumerate(cands):
        o = oracle32.icp(c.reading, c.ref_xyz, c.ref_nrm, c.T_init, **CHAIN)
        assert o["status"] == 0 and edges[k]["status"] == 0
        dt, dr = pose_error(o["T"], edges[k]["T_from_to"].reshape(4, 4))
        assert dt < TOL_TRANS and dr < TOL_ROT, (k, dt, dr)
        assert edges[k]["iterations"] == o["iterations"]
        assert edges[k]["overlap"] == pytest.approx(o["overlap"], rel=1e-12)
        res = oracle32.partial_chain(c.reading, c.ref_xyz, c.ref_nrm, o["T"], **CHAIN)
        assert edges[k]["residual"] == pytest.approx(res["residual"], rel=1e-3)
        expect = (not o["max_iter_reached"]) and o["overlap"] >= cfg.overlap_threshold and res["residual"] <= cfg.residual_error_threshold
        assert bool(edges[k]["accepted"]) == expect
        np.testing.assert_allclose(edges[k]["cov"].reshape(6, 6), o["cov"], rtol=1e-5, atol=1e-14)
