"""BASELINE.json configs[3] at test size: a loop-closing drive through the street grid of synth.city_route, run
through the C++ facade pgslam::PoseGraphSlam<float> (tools/slam_run), with a sample of its ICP calls -- scan-to-local-map
and loop closure -- recorded and replayed through the CPU oracle: same transform (1e-5 m / 1e-5 rad), same iteration
count.  The product itself has no CPU path; record / replay is how the oracle checks what ran inside the facade."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)


def test_city_route_revisits_streets():
    """Host logic of the generator: constant spacing, gentle turns, and streets driven twice."""
    from pgslam_amd import synth
    from scipy.spatial import cKDTree
    block, poses = synth.city_route(400, 1.5)
    P = np.array([p[:2, 3] for p in poses])
    d = np.linalg.norm(np.diff(P, axis=0), axis=1)
    assert abs(d.max() - 1.5) < 1e-6 and d.min() > 1.49
    pairs = cKDTree(P).query_pairs(3.0)
    assert sum(1 for a, b in pairs if abs(a - b) > 100) > 50
    odom = synth.city_odometry(poses)
    assert np.linalg.norm(odom[-1][:3, 3] - poses[-1][:3, 3]) > 0.05        # the odometry drifts
    city = synth.make_city(block)
    xyz, nrm = synth.make_city_scan(city, poses[10], 2000, 10)
    assert xyz.shape == (2000, 3) and np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_slam_facade_closes_loops_and_replays_through_the_oracle():
    import bench
    from oracle import Oracle
    seq = bench.build_sequence(400, 4000, 1.5)
    exe = bench.build_slam_run()
    rec = "/tmp/pgslam_amd_test_replay.bin"
    out = subprocess.run([exe, seq, "--record", "24", rec], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    # the facade did what configs[3] names: keyframes, loop closures, pose-graph solves, and it stayed on the road
    assert res["scans"] == 400 and res["keyframes"] >= 20
    assert res["loops_closed"] >= 1 and res["loop_edges"] == res["loops_closed"] and res["optimizer_runs"] >= 1
    assert res["tracking_error_max_m"] < 0.5 and res["keyframe_error_max_m"] < 0.5
    assert res["scans_not_converged"] <= 4
    recs = bench.read_replay(rec)
    assert len(recs) >= 16 and any(r["kind"] == 1 for r in recs)
    o = Oracle(np.float32)
    for r in recs:
        assert r["status"] == 0
        ref = o.icp(r["reading"], r["ref_xyz"], r["ref_nrm"], r["T_init"], **CHAIN)
        d = np.linalg.inv(ref["T"]) @ r["T_out"]
        dt = float(np.linalg.norm(d[:3, 3]))
        # T_out left the facade as a PM::Matrix<float>: the angle is taken from the skew part (sin of the angle), because
        # arccos(trace) turns the 6e-8 rounding of a float rotation matrix into 3e-4 rad
        dr = float(np.linalg.norm([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) / 2.0)
        assert dt < 1e-5 and dr < 1e-5, (r["kind"], r["scan"], dt, dr)
        assert ref["iterations"] == r["iterations"] and int(ref["converged"]) == r["converged"]


@pytest.mark.gpu
def test_probe_that_keeps_its_indexed_map_changes_nothing():
    """Localizer keeps the neighbour composition's world-frame map indexed between scans (the reference assembles and
    indexes it for every overlap check, Localizer.hpp:282-348).  With PGSLAM_PROBE_REBUILD set the facade does what the
    reference does: every figure of the run -- keyframes, loops, rebuilds, tracking errors to the printed digit -- is the same."""
    import bench
    import os
    seq = bench.build_sequence(400, 4000, 1.5)
    exe = bench.build_slam_run()
    runs = []
    for env in ({}, {"PGSLAM_PROBE_REBUILD": "1"}):
        out = subprocess.run([exe, seq], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    same = ("scans", "keyframes", "loop_edges", "loop_candidates_tried", "loops_closed", "optimizer_runs", "optimizer_iterations",
            "map_rebuilds", "mean_icp_iterations", "scans_not_converged", "tracking_error_rms_m", "tracking_error_max_m",
            "tracking_error_last_m", "keyframe_error_rms_m", "keyframe_error_max_m")
    for k in same:
        assert runs[0][k] == runs[1][k], (k, runs[0][k], runs[1][k])


@pytest.mark.gpu
def test_overlap_probe_seeded_from_the_icp_changes_nothing():
    """Round 6: the overlap probe's matcher (Localizer.hpp:210,231 -> 282-348, up to once per scan) is seeded with the correspondences
    the ICP of the same scan ended with, moved through the keyframes the two maps share (pgicp_partial_chain_seeded).  Seeds are
    candidates only: with PGSLAM_PROBE_SEEDS=0 every figure of the run -- keyframes, loops, rebuilds, tracking errors to the
    printed digit -- is the same, and the seeded run did seed its probes."""
    import bench
    import os
    seq = bench.build_sequence(300, 20000, 1.2)
    exe = bench.build_slam_run()
    runs = []
    for env in ({}, {"PGSLAM_PROBE_SEEDS": "0"}):
        out = subprocess.run([exe, seq, "--filters", "sensor"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    same = ("scans", "keyframes", "loop_edges", "loop_candidates_tried", "loops_closed", "optimizer_runs", "optimizer_iterations",
            "map_rebuilds", "mean_icp_iterations", "scans_not_converged", "tracking_error_rms_m", "tracking_error_max_m",
            "tracking_error_last_m", "keyframe_error_rms_m", "keyframe_error_max_m")
    for k in same:
        assert runs[0][k] == runs[1][k], (k, runs[0][k], runs[1][k])
    assert runs[0]["overlap_probes_seeded"] > 50 and runs[1]["overlap_probes_seeded"] == 0
