"""The localizer's input stage on the device (pgicp_filter_cloud): input_filters_.apply + the sensor transform
(/root/reference/src/pgslam/Localizer.hpp:103-106) in one pass.  The kept points must be those the oracle's restatement of
libpointmatcher's filters keeps (oracle/icp_oracle.c: orc_filter_chain, citing upstream's DataPointsFilters/*.cpp), index for
index; the moved coordinates those of the oracle's RigidTransformation; and the device copy must serve as an ICP reading."""
import numpy as np
import pytest

from pgslam_amd import icp, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_filter_chain_keeps_what_the_oracle_keeps(ctx, oracle32, oracle64, dtype):
    o = oracle32 if dtype == np.float32 else oracle64
    w = synth.make_two_scans(6000, rings=16)
    xyz, nrm = w["ref_xyz"].astype(dtype), w["ref_nrm"].astype(dtype)
    n = len(xyz)
    f = np.concatenate([xyz, np.ones((n, 1), dtype=dtype)], axis=1)
    f[17, 1] = np.nan
    f[4000, 3] = np.nan                                         # a NaN in the homogeneous row counts too
    obs = (np.array([0.1, 0.2, 1.7], dtype=dtype) - xyz).astype(dtype)
    d = np.concatenate([nrm, obs, np.arange(n, dtype=dtype)[:, None]], axis=1)     # normals | observationDirections | an id row
    T = synth.se3(x=0.4, y=-0.2, z=1.1, yaw=0.3, pitch=0.05, roll=-0.02)
    chains = [
        [(icp.FILTER_IDENTITY,)],
        [(icp.FILTER_REMOVE_NAN,), (icp.FILTER_MAX_DIST, 25.0), (icp.FILTER_MIN_DIST, 1.5)],
        [(icp.FILTER_BOUNDING_BOX, -2.0, -1.5, -3.0, 2.0, 1.5, 3.0, 1.0), (icp.FILTER_FIX_STEP, 3)],
        [(icp.FILTER_FIX_STEP, 2), (icp.FILTER_RANDOM_SAMPLING, 0.6, 7), (icp.FILTER_MAX_DIST, 30.0), (icp.FILTER_FIX_STEP, 5)],
        [(icp.FILTER_REMOVE_NAN,), (icp.FILTER_RANDOM_SAMPLING, 0.25, 123456789), (icp.FILTER_BOUNDING_BOX, -50, -50, -50, 50, 50, 50, 0.0)],
        # along one axis (dim = p[1] - 1), and a NaN in front of the distance filters: it fails both comparisons
        [(icp.FILTER_MAX_DIST, 12.0, 1), (icp.FILTER_MIN_DIST, -3.0, 2), (icp.FILTER_MIN_DIST, 0.2, 3)],
        [(icp.FILTER_MIN_DIST, 2.0)],
        [(icp.FILTER_MAX_DIST, -40.0)],                                       # |maxDist|, as upstream takes it
        [(icp.FILTER_MAX_POINT_COUNT, 1500, 11), (icp.FILTER_FIX_STEP, 2), (icp.FILTER_MAX_POINT_COUNT, 100000, 3)],
    ]
    for filters in chains:
        for Tm in (None, T):
            of, od, idx, dev = ctx.filter_cloud(filters, f, d, T=Tm, rotate_rows=(0, 3))
            want = o.filter_chain(filters, f)
            assert np.array_equal(idx, want), filters
            assert len(of) == len(want) and dev.n == len(want)
            if Tm is None:
                assert np.array_equal(of.view(np.uint8), f[want].view(np.uint8)) and np.array_equal(od.view(np.uint8), d[want].view(np.uint8))
            else:
                clean = ~np.any(np.isnan(f[want, :3]), axis=1)
                assert np.array_equal(of[clean, :3], o.transform(Tm, f[want][clean, :3]))        # RigidTransformation: R p + t
                assert np.array_equal(of[:, 3:].view(np.uint8), f[want][:, 3:].view(np.uint8))
                assert np.array_equal(od[:, 0:3], o.transform(Tm, d[want][:, 0:3], rotate_only=True))    # normals rotate
                assert np.array_equal(od[:, 3:6], o.transform(Tm, d[want][:, 3:6], rotate_only=True))
                assert np.array_equal(od[:, 6], d[want][:, 6])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_distance_filters_at_their_limits(ctx, oracle32, oracle64, dtype):
    """points whose norm lands exactly on, one ulp below and one ulp above the limit: the device computes the norm as upstream
    does -- sqrt((x x + y y) + z z) in T, correctly rounded -- and compares strictly"""
    o = oracle32 if dtype == np.float32 else oracle64
    rng = np.random.default_rng(5)
    d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lim = dtype(7.25)
    pts = (d * float(lim)).astype(dtype)                                       # norms within a few ulps of the limit, both sides
    pts[:4] = np.array([[3, 4, 0], [0, 5, 0], [5, 0, 0], [0, 3, 4]], dtype=dtype) * dtype(1.45)      # exactly 7.25
    f = np.concatenate([pts, np.ones((len(pts), 1), dtype=dtype)], axis=1)
    nrm = np.sqrt((pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2])
    assert (nrm == lim).sum() >= 4 and (nrm < lim).sum() > 1000 and (nrm > lim).sum() > 1000
    for filters in ([(icp.FILTER_MAX_DIST, float(lim))], [(icp.FILTER_MIN_DIST, float(lim))]):
        _, _, idx, _ = ctx.filter_cloud(filters, f, None)
        want = o.filter_chain(filters, f)
        assert np.array_equal(idx, want)
        keep = nrm < lim if filters[0][0] == icp.FILTER_MAX_DIST else nrm > lim
        assert np.array_equal(want, np.nonzero(keep)[0])                       # and numpy's own sqrt agrees with both


def test_fixstep_with_a_changing_step(oracle32):
    """[EXT] FixStepSampling.cpp: step *= stepMult after every cloud, clamped at endStep (the caller's state across calls)"""
    steps, s = [], 8.0
    for _ in range(5):
        steps.append(int(s))
        s = oracle32.fixstep_next(s, 8.0, 2.0, 0.5)
    assert steps == [8, 4, 2, 2, 2]
    s = 3.0
    seq = []
    for _ in range(4):
        seq.append(int(s))
        s = oracle32.fixstep_next(s, 3.0, 10.0, 1.5)
    assert seq == [3, 4, 6, 10]


def test_filtered_cloud_is_an_icp_reading_without_a_second_upload(ctx, oracle32):
    """the device copy of the filtered, transformed scan aligns like the host copy of it"""
    chain = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
    w = synth.make_scan_to_map(n_scan=6000, n_map=40_000, n_queries=1, n_map_poses=4, rings=16)
    ctx.set_params(**chain)
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    T_rs = synth.se3(x=0.3, y=-0.1, z=0.2, yaw=0.1)
    sensor = oracle32.transform(synth.se3_inv(T_rs), w.scans_xyz[0])             # the scan as the sensor sees it
    f = np.concatenate([sensor, np.ones((len(sensor), 1), dtype=np.float32)], axis=1)
    filters = [(icp.FILTER_MAX_DIST, 60.0), (icp.FILTER_FIX_STEP, 2)]
    of, _, idx, dev = ctx.filter_cloud(filters, f, None, T=T_rs)
    Ta, sa = ctx.align(mid, dev, w.T_init[0])
    Tb, sb = ctx.align(mid, np.ascontiguousarray(of[:, :3]), w.T_init[0])
    assert np.array_equal(Ta, Tb) and sa["iterations"] == sb["iterations"] and sa["n_kept"] == sb["n_kept"]
    r = oracle32.icp(of[:, :3], w.map_xyz, w.map_nrm, w.T_init[0], **chain)
    d = np.linalg.inv(r["T"]) @ Ta
    assert np.linalg.norm(d[:3, 3]) < 1e-5 and sa["iterations"] == r["iterations"]
    ctx.destroy_map(mid)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_only_pass_lists_the_dropped_points(ctx, oracle32, oracle64, dtype):
    """pgicp_filter_cloud_dev: the caller's arrays stay as they are; the kept count, the ascending list of dropped indices (the
    complement of what the oracle keeps) and a device copy that aligns like the compacted host cloud"""
    o = oracle32 if dtype == np.float32 else oracle64
    chain = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
    w = synth.make_scan_to_map(n_scan=6000, n_map=40_000, n_queries=1, n_map_poses=4, rings=16)
    f = np.concatenate([w.scans_xyz[0], np.ones((len(w.scans_xyz[0]), 1), dtype=np.float32)], axis=1).astype(dtype)
    f[11, 0] = np.nan
    before = f.copy()
    for filters in ([(icp.FILTER_REMOVE_NAN,), (icp.FILTER_MAX_DIST, 35.0), (icp.FILTER_BOUNDING_BOX, -1.2, -0.9, -2.0, 1.2, 0.9, 0.5, 1.0)],
                    [(icp.FILTER_IDENTITY,)],
                    [(icp.FILTER_FIX_STEP, 2)]):                                  # (half the cloud dropped: more than the list holds)
        kept, dev, dropped = ctx.filter_cloud_dev(filters, f, dropped_cap=1024)
        want = o.filter_chain(filters, f)
        assert kept == len(want) and np.array_equal(f.view(np.uint8), before.view(np.uint8))
        if len(f) - kept > 1024:
            assert dropped is None
            continue
        assert np.array_equal(dropped, np.setdiff1d(np.arange(len(f)), want))
        ctx.set_params(**chain)
        mid = ctx.set_map(w.map_xyz.astype(dtype), w.map_nrm.astype(dtype), center=True, dtype=dtype)
        Ta, sa = ctx.align(mid, dev, w.T_init[0], dtype=dtype)
        Tb, sb = ctx.align(mid, np.ascontiguousarray(f[want][:, :3]), w.T_init[0], dtype=dtype)
        assert np.array_equal(Ta, Tb) and sa["iterations"] == sb["iterations"] and sa["n_kept"] == sb["n_kept"]
        ctx.destroy_map(mid)


def test_local_map_from_device_resident_keyframes_equals_the_host_flow(ctx, oracle32):
    """Keyframe clouds kept in device memory of the caller's own (pgicp_device_alloc / _copy), the map assembled from them
    there (pgicp_build_local_map, mem = DEVICE) and indexed in place (pgicp_map_create, mem = DEVICE): the same cloud and the
    same alignment as through the host (LocalMap.hpp:209-224 + setMap), bit for bit -- what the C++ facade's localizer does."""
    chain = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
    ctx.set_params(**chain)
    w = synth.make_scan_to_map(n_scan=5000, n_map=30_000, n_queries=1, n_map_poses=3, rings=16)
    third = len(w.map_xyz) // 3
    kx = [w.map_xyz[i * third:(i + 1) * third] for i in range(3)]
    kn = [w.map_nrm[i * third:(i + 1) * third] for i in range(3)]
    Ts = [np.eye(4), synth.se3(x=0.4, y=-0.2, yaw=0.05), synth.se3(x=-0.3, z=0.1, yaw=-0.02)]
    hx, hn = ctx.build_local_map(kx, kn, Ts)                                  # through the host
    dx = [ctx.device_cloud(c) for c in kx]
    dn = [ctx.device_cloud(c) for c in kn]
    mx, mn = ctx.build_local_map(dx, dn, Ts)                                  # in device memory
    assert np.array_equal(ctx.device_download(mx), hx) and np.array_equal(ctx.device_download(mn), hn)
    ox, on = oracle32.build_local_map(kx, kn, Ts)
    assert np.array_equal(hx, ox) and np.array_equal(hn, on)
    mid_h = ctx.set_map(hx, hn, center=True)
    mid_d = ctx.set_map(mx, mn, center=True)
    Th, sh = ctx.align(mid_h, w.scans_xyz[0], w.T_init[0])
    Td, sd = ctx.align(mid_d, w.scans_xyz[0], w.T_init[0])
    assert np.array_equal(Th, Td) and sh["iterations"] == sd["iterations"] and sh["n_kept"] == sd["n_kept"] and sh["residual"] == sd["residual"]
    for d in dx + dn + [mx, mn]:
        ctx.device_free(d)
    ctx.destroy_map(mid_h)
    ctx.destroy_map(mid_d)


def test_filter_chain_edge_cases(ctx):
    """a chain that keeps nothing, a one-point cloud, a chain of the maximum length, refused arguments"""
    f = np.array([[0.1, 0.2, 0.3, 1.0], [5.0, 0.0, 0.0, 1.0], [0.0, 9.0, 0.0, 1.0]], dtype=np.float32)
    # everything inside the box, removeInside: nothing is left
    of, od, idx, dev = ctx.filter_cloud([(icp.FILTER_BOUNDING_BOX, -20, -20, -20, 20, 20, 20, 1.0)], f, None)
    assert len(idx) == 0 and of.shape[0] == 0
    # one point, kept / dropped
    of, od, idx, dev = ctx.filter_cloud([(icp.FILTER_MAX_DIST, 1.0)], f[:1], None)
    assert list(idx) == [0] and np.array_equal(of[0, :3], f[0, :3])
    of, od, idx, dev = ctx.filter_cloud([(icp.FILTER_MIN_DIST, 1.0)], f[:1], None)
    assert len(idx) == 0
    # the longest chain the ABI takes (eight filters), each a no-op here
    of, od, idx, dev = ctx.filter_cloud([(icp.FILTER_IDENTITY,)] * 4 + [(icp.FILTER_MAX_DIST, 100.0)] * 4, f, None)
    assert list(idx) == [0, 1, 2]
    with pytest.raises(icp.PgicpError):
        ctx.filter_cloud([(icp.FILTER_IDENTITY,)] * 9, f, None)                      # more than PGICP_MAX_FILTERS
    with pytest.raises(icp.PgicpError):
        ctx.filter_cloud([(99,)], f, None)                                           # unknown filter type
    with pytest.raises(icp.PgicpError):
        ctx.filter_cloud([(icp.FILTER_FIX_STEP, 0)], f, None)                        # a step below one


def test_device_memory_calls_refuse_bad_arguments(ctx):
    import ctypes as C
    p = C.c_void_p()
    assert ctx.lib.pgicp_device_alloc(ctx.h, C.c_size_t(0), C.byref(p)) == icp.ERR_ARG
    d = ctx.device_empty(4, 3, np.float32)
    host = np.zeros(12, dtype=np.float32)
    assert ctx.lib.pgicp_device_copy(ctx.h, C.c_void_p(d.ptr), C.c_void_p(host.ctypes.data), C.c_size_t(48), C.c_int(7)) == icp.ERR_ARG
    assert ctx.lib.pgicp_device_copy(ctx.h, C.c_void_p(d.ptr), None, C.c_size_t(48), C.c_int(0)) == icp.ERR_ARG
    assert ctx.lib.pgicp_device_copy(ctx.h, C.c_void_p(d.ptr), C.c_void_p(host.ctypes.data), C.c_size_t(0), C.c_int(0)) == icp.OK
    host[:] = np.arange(12)
    assert ctx.lib.pgicp_device_copy(ctx.h, C.c_void_p(d.ptr), C.c_void_p(host.ctypes.data), C.c_size_t(48), C.c_int(0)) == icp.OK
    assert np.array_equal(ctx.device_download(d).ravel(), host)
    ctx.device_free(d)
    assert ctx.lib.pgicp_device_free(None, None) == icp.OK                            # nothing to free, no context: fine


# ---------------------------------------------------------------- round 6: densities, MaxDensity, SamplingSurfaceNormal in a chain
def _write_cloud(path, xyz, dtype):
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("i", len(xyz)))
        f.write(np.ascontiguousarray(xyz, dtype=dtype).tobytes())


def test_densities_and_max_density_filter_through_the_yaml_loader(tmp_path, oracle32):
    """SurfaceNormalDataPointsFilter{keepDensities: 1} + MaxDensityDataPointsFilter{maxDensity} as a user's input-filter file names
    them (Localizer.hpp:73-78): the neighbour search runs on the device (pgicp_surface_normals), densities and the thinning on the
    host.  Against the oracle: the same neighbours, the same densities (in T, bit for bit), the same kept points."""
    from test_filters_host import apply_filters
    s = synth.make_two_scans(6000, rings=16)
    xyz = s["ref_xyz"].astype(np.float32)
    sn = oracle32.surface_normals(xyz, 10)
    dens = oracle32.densities(xyz, sn["ids"])
    md = float(np.median(dens))
    yaml = ("- SurfaceNormalDataPointsFilter:\n    knn: 10\n    keepDensities: 1\n- MaxDensityDataPointsFilter:\n    maxDensity: %.9g\n    seed: 11\n" % md)
    pts, nrm = apply_filters(tmp_path, yaml, xyz, np.float32)
    keep = oracle32.max_density_keep(dens, np.float32(md), seed=11)
    assert 0.5 * len(xyz) < keep.sum() < 0.95 * len(xyz)
    assert len(pts) == keep.sum() and np.array_equal(pts, xyz[keep])
    assert nrm is not None and np.all(np.abs(np.sum(nrm * sn["normals"][keep], axis=1)) > 1 - 1e-4)


@pytest.mark.parametrize("chain", ["yaml", "default"])
def test_icp_whose_reference_filter_is_sampling_surface_normal(tmp_path, oracle32, chain):
    """A chain that GETS its normals the way upstream's default chain does -- SamplingSurfaceNormalDataPointsFilter on the
    reference (referenceDataPointsFilters, Localizer.hpp:314-315; ICP::operator(), LoopCloser.hpp:98) -- through loadFromYaml /
    setDefault of the shim, against the oracle run on the oracle's own filtered reference."""
    import struct
    import subprocess
    from test_cpp_dropin import build
    s = synth.make_two_scans(8000, rings=16)
    rd, ref = s["reading_xyz"].astype(np.float32), s["ref_xyz"].astype(np.float32)
    if chain == "yaml":
        yaml = ("referenceDataPointsFilters:\n  - SamplingSurfaceNormalDataPointsFilter:\n      ratio: 0.6\n      knn: 10\n      samplingMethod: 1\n"
                "matcher:\n  KDTreeMatcher:\n    knn: 1\n    maxDist: 2.0\n"
                "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.85\n"
                "errorMinimizer:\n  PointToPlaneWithCovErrorMinimizer:\n    sensorStdDev: 0.01\n"
                "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
                "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n")
        fy = str(tmp_path / "chain.yaml")
        open(fy, "w").write(yaml)
        f = oracle32.sampling_surface_normal(ref, knn=10, ratio=0.6, sampling_method=1)
        oreading, okw = rd, dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
    else:
        # upstream's defaults: reading RandomSampling 0.75 (seed 1), reference SamplingSurfaceNormal (0.5, knn 7, method 0, seed 1),
        # KDTree maxDist inf, TrimmedDist 0.85, PointToPlane, Counter 40, Differential (0.001, 0.001, 3)
        fy = "default"
        f = oracle32.sampling_surface_normal(ref, knn=7, ratio=0.5, sampling_method=0, seed=1)
        kept = oracle32.filter_chain([(6, 0.75, 1)], rd)
        oreading = rd[kept]
        okw = dict(max_dist=float("inf"), trim_ratio=0.85, max_iters=40, min_diff_rot=0.001, min_diff_trans=0.001, smooth_length=3, sensor_std_dev=0.01)
    k = f["keep"]
    fr, fi, ft, fo = (str(tmp_path / n) for n in ("reading.bin", "ref.bin", "tinit.bin", "out.bin"))
    _write_cloud(fr, rd, np.float32)
    _write_cloud(fi, ref, np.float32)
    open(ft, "wb").write(np.ascontiguousarray(s["T_init"], dtype=np.float64).tobytes())
    r = subprocess.run([build("icp_apply"), "f32", fy, fr, fi, ft, fo], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    raw = open(fo, "rb").read()
    T = np.frombuffer(raw, dtype=np.float64, count=16).reshape(4, 4)
    m, hn = struct.unpack("ii", raw[128:136])
    assert m == k.sum() and hn == 1
    o = oracle32.icp(oreading, f["xyz"][k], f["normals"][k], s["T_init"], **okw)
    assert o["status"] == 0
    d = np.linalg.inv(o["T"]) @ T
    dt = np.linalg.norm(d[:3, 3])
    dr = np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))
    # the two sides' box normals agree to float rounding (the same Jacobi on the same sums), not bit for bit: 1e-5 m / 1e-5 rad
    assert dt < 1e-5 and dr < 1e-5, (dt, dr)
    truth = np.linalg.inv(s["T_truth"]) @ T
    assert np.linalg.norm(truth[:3, 3]) < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("minimizer", ["PointToPoint", "PointToPlane"])
def test_get_overlap_of_a_reading_with_simple_sensor_noise(tmp_path, oracle32, minimizer):
    """getOverlap() as pgslam reads it (Localizer.hpp:278, LoopCloser.hpp:331) when the user's chain gives the reading a
    `simpleSensorNoise` descriptor (readingDataPointsFilters: SimpleSensorNoiseDataPointsFilter; the point-to-plane minimizer also wants
    `normals` on the reading): the share of the LAST error elements whose distance lies below mean + noise, not the weighted ratio.
    Shim (loadFromYaml -> ICP::operator() -> pgicp_debug_last_matches) against the oracle's ICP, its last correspondences and
    orc_sensor_noise_overlap."""
    import struct
    import subprocess
    from test_cpp_dropin import build
    s = synth.make_two_scans(8000, rings=16)
    rd, ref = s["reading_xyz"].astype(np.float32), s["ref_xyz"].astype(np.float32)
    p2p = minimizer == "PointToPoint"
    yaml = ("readingDataPointsFilters:\n"
            + ("" if p2p else "  - SurfaceNormalDataPointsFilter:\n      knn: 8\n")
            + "  - SimpleSensorNoiseDataPointsFilter:\n      sensorType: 0\n      gain: 1\n"
            "referenceDataPointsFilters:\n  - SurfaceNormalDataPointsFilter:\n      knn: 8\n"
            "matcher:\n  KDTreeMatcher:\n    knn: 1\n    maxDist: 2.0\n"
            "outlierFilters:\n  - TrimmedDistOutlierFilter:\n      ratio: 0.8\n"
            "errorMinimizer:\n  %sErrorMinimizer:\n    sensorStdDev: 0.01\n" % minimizer +
            "transformationCheckers:\n  - CounterTransformationChecker:\n      maxIterationCount: 30\n"
            "  - DifferentialTransformationChecker:\n      minDiffRotErr: 0.001\n      minDiffTransErr: 0.01\n      smoothLength: 3\n")
    fy, fr, fi, ft, fo = (str(tmp_path / n) for n in ("chain.yaml", "reading.bin", "ref.bin", "tinit.bin", "out.bin"))
    open(fy, "w").write(yaml)
    _write_cloud(fr, rd, np.float32)
    _write_cloud(fi, ref, np.float32)
    open(ft, "wb").write(np.ascontiguousarray(s["T_init"], dtype=np.float64).tobytes())
    r = subprocess.run([build("icp_apply"), "f32", fy, fr, fi, ft, fo], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    raw = open(fo, "rb").read()
    T = np.frombuffer(raw, dtype=np.float64, count=16).reshape(4, 4)
    overlap, ratio = struct.unpack("dd", raw[136:152])
    nrm = oracle32.surface_normals(ref, knn=8)["normals"] if hasattr(oracle32, "surface_normals") else None
    assert nrm is not None
    o = oracle32.icp(rd, ref, nrm, s["T_init"], max_dist=2.0, trim_ratio=0.8, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
                     smooth_length=3, error_minimizer=(1 if p2p else 0))
    assert o["status"] == 0
    d = np.linalg.inv(o["T"]) @ T
    assert np.linalg.norm(d[:3, 3]) < 1e-4
    noise = oracle32.simple_sensor_noise(rd, 0, 1.0)
    w = ((o["last_ids"] >= 0) & (o["last_d2"] <= np.float32(o["trim_limit"]))).astype(np.float32)
    want = oracle32.sensor_noise_overlap(o["last_d2"], w, noise)
    assert ratio == pytest.approx(0.8, abs=1e-3)
    assert abs(overlap - ratio) > 0.01                       # the branch was taken: another quantity altogether
    assert overlap == pytest.approx(want, abs=2e-3)           # (normals and sums agree to float rounding between the two sides: a few pairs near the bound)
