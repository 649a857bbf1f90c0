"""The host-side data-point filters of the drop-in's PointMatcher shim (include/pgslam_amd/pointmatcher.hpp) against the
oracle's statement of the same upstream filters, through the YAML loader a pgslam user's file goes through
(input_filters_ / referenceDataPointsFilters, /root/reference/src/pgslam/Localizer.hpp:73-78, 103, 314-315).  No device:
SamplingSurfaceNormalDataPointsFilter runs on the host (once per keyframe / map)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from pgslam_amd import synth
from test_cpp_dropin import build


def apply_filters(tmp_path, yaml, xyz, dtype, want_noise=False):
    exe = build("filter_apply")
    fy, fi, fo = (str(tmp_path / n) for n in ("f.yaml", "in.bin", "out.bin"))
    open(fy, "w").write(yaml)
    with open(fi, "wb") as f:
        f.write(struct.pack("i", len(xyz)))
        f.write(np.ascontiguousarray(xyz, dtype=dtype).tobytes())
    r = subprocess.run([exe, "f32" if dtype == np.float32 else "f64", fy, fi, fo], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(fo, "rb").read()
    m, hn, hd = struct.unpack("iii", raw[:12])
    it = np.dtype(dtype).itemsize
    pts = np.frombuffer(raw, dtype=dtype, count=3 * m, offset=12).reshape(m, 3)
    nrm = np.frombuffer(raw, dtype=dtype, count=3 * m, offset=12 + 3 * m * it).reshape(m, 3) if hn else None
    if want_noise:
        off = 12 + 3 * m * it * (2 if hn else 1) + (m * it if hd else 0)
        hs, = struct.unpack("i", raw[off:off + 4])
        return pts, nrm, (np.frombuffer(raw, dtype=dtype, count=m, offset=off + 4) if hs else None)
    return pts, nrm


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("method", [0, 1])
def test_sampling_surface_normal_filter_equals_the_oracle(tmp_path, oracle32, oracle64, dtype, method):
    o = oracle32 if dtype == np.float32 else oracle64
    s = synth.make_two_scans(7000, rings=16)
    xyz = s["ref_xyz"].astype(dtype)
    yaml = ("- SamplingSurfaceNormalDataPointsFilter:\n    ratio: 0.4\n    knn: 9\n    samplingMethod: %d\n    maxBoxDim: 1.5\n    seed: 23\n" % method)
    pts, nrm = apply_filters(tmp_path, yaml, xyz, dtype)
    r = o.sampling_surface_normal(xyz, knn=9, ratio=0.4, sampling_method=method, max_box_dim=1.5, seed=23)
    k = r["keep"]
    assert len(pts) == k.sum() and 0 < len(pts) < len(xyz)
    assert np.array_equal(pts, r["xyz"][k])                       # the same points, in cloud order (method 1: the same box means, bit for bit)
    # the same Jacobi on the same T-accumulated scatter: the normals agree to rounding (and in sign)
    np.testing.assert_allclose(nrm, r["normals"][k], atol=5e-6 if dtype == np.float32 else 1e-13)
    if method == 0:
        assert 0.3 < len(pts) / len(xyz) < 0.5                    # about `ratio` of the points of the boxes that were fused


def test_set_default_installs_upstreams_default_filters():
    """[EXT] ICPChaineBase::setDefault: RandomSampling on the reading, SamplingSurfaceNormal on the reference"""
    out = subprocess.run([build("test_dropin_cpu")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "setDefault filters ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("sensor", [0, 1, 2, 3, 4])
def test_simple_sensor_noise_filter_equals_the_oracle(tmp_path, oracle32, oracle64, dtype, sensor):
    """[EXT] SimpleSensorNoiseDataPointsFilter{sensorType, gain} through the YAML loader: the `simpleSensorNoise` descriptor the
    sensor-noise branch of getOverlap() reads (Localizer.hpp:278, LoopCloser.hpp:331), bit for bit the oracle's"""
    o = oracle32 if dtype == np.float32 else oracle64
    xyz = synth.make_two_scans(3000, rings=16)["ref_xyz"].astype(dtype)
    yaml = "- SimpleSensorNoiseDataPointsFilter:\n    sensorType: %d\n    gain: 1.5\n" % sensor
    pts, nrm, noise = apply_filters(tmp_path, yaml, xyz, dtype, want_noise=True)
    assert np.array_equal(pts, xyz) and nrm is None and noise is not None
    ref = o.simple_sensor_noise(xyz, sensor, 1.5)
    assert np.array_equal(noise, ref)
    r = np.linalg.norm(xyz.astype(np.float64), axis=1)
    if sensor == 3:
        np.testing.assert_allclose(noise, 1.5 * r * r * 0.5 * 0.00285, rtol=2e-6 if dtype == np.float32 else 1e-12)
    else:
        a, b, c = {0: (0.012, 0.0068, 0.0008), 1: (0.028, 0.0013, 0.0001), 2: (0.018, 0.0006, 0.0015), 4: (0.004, 0.0053, -0.0092)}[sensor]
        np.testing.assert_allclose(noise, 1.5 * np.maximum(a, b * r + c), rtol=2e-6 if dtype == np.float32 else 1e-12)


def test_simple_sensor_noise_filter_refuses_what_it_does_not_know(tmp_path):
    xyz = np.zeros((4, 3), np.float32)
    for bad in ("    sensorType: 5\n", "    sensorType: 1.5\n", "    gain: 0.5\n"):
        with pytest.raises(AssertionError):
            apply_filters(tmp_path, "- SimpleSensorNoiseDataPointsFilter:\n" + bad, xyz, np.float32)
