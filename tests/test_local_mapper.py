"""Streaming local mapper: host logic against the CPU oracle here,
GPU parity in the gpu-marked tests."""
import numpy as np
import pytest

from pgslam_amd import synth
from pgslam_amd.local_mapper import Keyframe, LocalMapperConfig, StreamingLocalMapper, composition_transforms

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=1e-3, min_diff_trans=1e-2, smooth_length=3,
             sensor_std_dev=0.01)


class OracleBackend:
    """The mapper's backend interface on top of the CPU oracle (test infrastructure only)."""

    def __init__(self, oracle):
        self.o = oracle
        self.maps = {}
        self.next = 0

    def set_params(self, **kw):
        pass

    def build_local_map(self, xs, ns, Ts):
        return self.o.build_local_map(xs, ns, Ts)

    def set_map(self, xyz, nrm, center=True):
        self.maps[self.next] = self.o.map_create(xyz, nrm, center=center)
        self.next += 1
        return self.next - 1

    def destroy_map(self, m):
        self.o.map_free(self.maps.pop(m))

    def align(self, m, reading, T_init):
        r = self.o.icp_map(self.maps[m], reading, T_init, **CHAIN)
        assert r["status"] == 0
        return r["T"], r


@pytest.fixture(scope="module")
def drive():
    return synth.make_drive(16, n_pts=4000, step=1.6, rings=16)


def pose_err(A, B):
    dT = np.linalg.inv(A) @ B
    return np.linalg.norm(dT[:3, 3]), np.arccos(np.clip((np.trace(dT[:3, :3]) - 1) / 2, -1, 1))


def test_composition_order_is_reference_then_newest_to_oldest():
    ks = [Keyframe(i, None, None, synth.se3(x=float(i))) for i in range(4)]
    order, Ts = composition_transforms(ks)
    assert [k.id for k in order] == [3, 2, 1, 0]                       # LocalMap.hpp:213-223
    np.testing.assert_allclose(Ts[0], np.eye(4))
    np.testing.assert_allclose(Ts[1][:3, 3], [-1.0, 0, 0], atol=1e-12)  # T_refkf_world * T_world_kf
    np.testing.assert_allclose(Ts[3][:3, 3], [-3.0, 0, 0], atol=1e-12)


def test_mapper_tracks_the_drive_and_slides_the_window(oracle32, drive):
    m = StreamingLocalMapper(OracleBackend(oracle32), LocalMapperConfig(capacity=3, overlap_threshold=0.8))
    odom_err, icp_err = [], []
    for s in range(len(drive.odom)):
        T = m.process(drive.odom[s], drive.scans_xyz[s], drive.scans_nrm[s])
        # poses are reported relative to the first keyframe's odometry pose
        T_true = drive.poses_true[s]
        icp_err.append(pose_err(T_true, T)[0])
        odom_err.append(pose_err(T_true, drive.odom[s])[0])
    assert m.keyframe_scans[0] == 0 and len(m.keyframe_scans) >= 4     # overlap dropped below 0.8 several times
    assert len(m.window) == 3 and m.rebuilds >= len(m.keyframe_scans)
    assert [k.id for k in m.window] == [m.next_kf_id - 3, m.next_kf_id - 2, m.next_kf_id - 1]   # the oldest dropped out
    # ICP keeps the error bounded (sparse 4k-point scans: centimetres) where raw odometry keeps drifting
    assert max(icp_err) < 0.15 and icp_err[-1] < odom_err[-1]
    m.close()


def test_reference_follows_the_closest_keyframe(oracle32, drive):
    """Case #2 of UpdateAfterIcp (Localizer.hpp:213-221): driving back towards an older keyframe
    swaps it into the reference slot without changing the keyframe set."""
    m = StreamingLocalMapper(OracleBackend(oracle32), LocalMapperConfig(capacity=3, overlap_threshold=0.8))
    fwd = list(range(0, 8))
    back = [6, 5, 4, 3, 2]
    odom = [drive.poses_true[s] for s in fwd + back]                    # perfect odometry: isolates the policy
    for k, s in enumerate(fwd + back):
        m.process(odom[k], drive.scans_xyz[s], drive.scans_nrm[s])
        ref = m.window[-1]
        d_ref = np.linalg.norm(ref.T_world_kf[:3, 3] - m.T_world_robot[:3, 3])
        if m.last_stats is not None and m.last_stats["overlap"] >= 0.8:
            assert all(d_ref <= np.linalg.norm(kf.T_world_kf[:3, 3] - m.T_world_robot[:3, 3]) + 1e-9 for kf in m.window)
    m.close()


@pytest.mark.gpu
def test_gpu_mapper_matches_oracle_mapper(oracle32, drive):
    import torch
    from pgslam_amd import icp
    dev = torch.device("cuda", 0)
    ctx = icp.Context(0, **CHAIN)
    cfg = LocalMapperConfig(capacity=3, overlap_threshold=0.8, chain=CHAIN)
    g = StreamingLocalMapper(ctx, cfg, to_device=lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    o = StreamingLocalMapper(OracleBackend(oracle32), LocalMapperConfig(capacity=3, overlap_threshold=0.8))
    for s in range(len(drive.odom)):
        Tg = g.process(drive.odom[s], drive.scans_xyz[s], drive.scans_nrm[s])
        To = o.process(drive.odom[s], drive.scans_xyz[s], drive.scans_nrm[s])
        dt, dr = pose_err(To, Tg)
        assert dt < 1e-4 and dr < 1e-4, (s, dt, dr)                      # errors may compound over the drive
    assert g.keyframe_scans == o.keyframe_scans and [k.id for k in g.window] == [k.id for k in o.window]
    # the device-resident map equals the oracle's assembly bit for bit
    order, Ts = composition_transforms(list(g.window))
    gx, gn = ctx.build_local_map([k.xyz for k in order], [k.nrm for k in order], Ts)
    ox, on = oracle32.build_local_map([k.xyz.cpu().numpy() for k in order], [k.nrm.cpu().numpy() for k in order], Ts)
    np.testing.assert_array_equal(gx.cpu().numpy(), ox)
    np.testing.assert_array_equal(gn.cpu().numpy(), on)
    g.close(); o.close(); ctx.close()


@pytest.mark.gpu
def test_gpu_async_rebuild_tracks_like_sync(drive):
    import torch
    from pgslam_amd import icp
    dev = torch.device("cuda", 0)
    up = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = {}
    for mode in (False, True):
        ctx = icp.Context(0, **CHAIN)
        builder = icp.Context(0, **CHAIN) if mode else None
        m = StreamingLocalMapper(ctx, LocalMapperConfig(capacity=3, overlap_threshold=0.8, chain=CHAIN, async_rebuild=mode),
                                 builder=builder, to_device=up)
        res[mode] = [m.process(drive.odom[s], drive.scans_xyz[s], drive.scans_nrm[s]) for s in range(len(drive.odom))]
        assert m.rebuilds >= 2
        m.close(); ctx.close()
        if builder:
            builder.close()
    for s, (A, B) in enumerate(zip(res[False], res[True])):
        dt, dr = pose_err(A, B)
        # which scan first sees the new map depends on thread timing; either way the pose is the same within
        # the accuracy ICP has on 4k-point scans (centimetres)
        assert dt < 0.15 and dr < 0.03, (s, dt, dr)
    for s, T in enumerate(res[True]):
        assert pose_err(drive.poses_true[s], T)[0] < 0.15


@pytest.mark.gpu
def test_gpu_fleet_equals_vehicles_stepped_alone(drive):
    """StreamingFleet shares only the ICP calls: every vehicle ends where it would have ended alone."""
    import torch
    from pgslam_amd import icp
    from pgslam_amd.local_mapper import StreamingFleet
    dev = torch.device("cuda", 0)
    up = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ctx = icp.Context(0, **CHAIN)
    cfg = LocalMapperConfig(capacity=3, overlap_threshold=0.8, chain=CHAIN)
    S = len(drive.odom)
    # vehicle v starts v scans into the drive (different maps, different keyframe times)
    starts = [0, 2, 5]
    fleet = StreamingFleet(ctx, len(starts), cfg, to_device=up)
    solo = [StreamingLocalMapper(ctx, cfg, to_device=up) for _ in starts]
    for t in range(S - max(starts)):
        idx = [s + t for s in starts]
        Tf = fleet.step([drive.odom[i] for i in idx], [drive.scans_xyz[i] for i in idx], [drive.scans_nrm[i] for i in idx])
        for v, i in enumerate(idx):
            Ts = solo[v].process(drive.odom[i], drive.scans_xyz[i], drive.scans_nrm[i])
            np.testing.assert_allclose(Tf[v], Ts, rtol=0, atol=1e-9)       # batch vs single: equal to rounding
    for v in range(len(starts)):
        assert fleet.mappers[v].keyframe_scans == solo[v].keyframe_scans
    fleet.close()
    for m in solo:
        m.close()
    ctx.close()


@pytest.mark.gpu
def test_gpu_host_scans_one_step_ahead_equal_device_resident(drive):
    """The caller owns HOST scans (Localizer.hpp:103-126): scan k + 1 travels on the copy stream (pgicp_upload_*) while
    scan k aligns (LocalizerMT.hpp:27-40).  Poses, keyframe decisions and the keyframes' device clouds are those of the
    run whose scans were resident in HBM beforehand -- pinned or pageable sources alike."""
    import torch
    from pgslam_amd import icp
    dev = torch.device("cuda", 0)
    S = len(drive.odom)
    cfg = LocalMapperConfig(capacity=3, overlap_threshold=0.8, chain=CHAIN)
    ctx = icp.Context(0, **CHAIN)
    resident = StreamingLocalMapper(ctx, cfg)
    want = [resident.process(drive.odom[s], torch.from_numpy(np.ascontiguousarray(drive.scans_xyz[s])).to(dev),
                             torch.from_numpy(np.ascontiguousarray(drive.scans_nrm[s])).to(dev)) for s in range(S)]
    for pinned in (False, True):
        if pinned:
            block = ctx.host_alloc((S,) + drive.scans_xyz[0].shape, np.float32)
            for s in range(S):
                block[s] = drive.scans_xyz[s]
            src = [block[s] for s in range(S)]
        else:
            src = [np.ascontiguousarray(x) for x in drive.scans_xyz]
        m = StreamingLocalMapper(ctx, cfg)
        m.pinned_sources = pinned
        got = [m.process(drive.odom[s], src[s], drive.scans_nrm[s], next_xyz=src[s + 1] if s + 1 < S else None) for s in range(S)]
        for s in range(S):
            np.testing.assert_array_equal(got[s], want[s])
        assert m.keyframe_scans == resident.keyframe_scans and m.rebuilds == resident.rebuilds
        for a, b in zip(m.window, resident.window):
            assert torch.equal(a.xyz, b.xyz) and torch.equal(a.nrm, b.nrm)
        m.close()
        if pinned:
            ctx.host_free(block)
    resident.close()
    ctx.close()
