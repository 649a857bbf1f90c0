"""ctypes binding of the CPU oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under pgslam_amd/ does.  See icp_oracle.c for the citations
of the reference call sites each function restates, and for the "parity
unpinned" statement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


VARIANTS = ("accum_t", "tie_high", "fma", "all3")


def build(force: bool = False, variant: str = "") -> str:
    """variant: "" = the oracle; "accum_t" / "tie_high" / "fma" / "all3" = the sensitivity builds of icp_oracle.c's header
    (one unverifiable assumption flipped each; used by tests/test_sensitivity.py and tools/sensitivity_envelope.py only)."""
    src = os.path.join(_HERE, "icp_oracle.c")
    name = "liboracle.so" if not variant else f"liboracle_{variant}.so"
    path = os.path.join(_HERE, name)
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
    return path


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("converged", C.c_int),
                ("max_iter_reached", C.c_int), ("overlap", C.c_double), ("residual", C.c_double),
                ("trim_limit", C.c_double), ("n_kept", C.c_int), ("n_finite", C.c_int),
                ("cov", C.c_double * 36)]


def _params_type(real):
    class Params(C.Structure):
        _fields_ = [("max_dist", real), ("trim_ratio", real), ("max_iters", C.c_int),
                    ("min_diff_rot", C.c_double), ("min_diff_trans", C.c_double),
                    ("smooth_length", C.c_int), ("sensor_std_dev", C.c_double),
                    ("use_kdtree", C.c_int), ("center_reference", C.c_int), ("outlier_max_dist", real),
                    ("quantile_scale", real), ("knn", C.c_int), ("minimizer", C.c_int), ("bound_max_rot", C.c_double),
                    ("bound_max_trans", C.c_double), ("normal_max_angle", real), ("robust_fct", C.c_int), ("robust_tuning", real),
                    ("robust_scale", C.c_int), ("robust_approx", real), ("pair_order", C.c_void_p)]
    return Params


class Checker(C.Structure):
    _fields_ = [("count", C.c_int), ("max_iters", C.c_int), ("smooth", C.c_int),
                ("min_rot", C.c_double), ("min_trans", C.c_double), ("n_hist", C.c_int),
                ("quat", C.c_double * (64 * 4)), ("trans", C.c_double * (64 * 3)),
                ("bound_rot", C.c_double), ("bound_trans", C.c_double)]


DEFAULT_CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001,
                     min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)


class Oracle:
    """dtype = np.float32 (PointMatcher<float>) or np.float64 (PointMatcher<double>)."""

    def __init__(self, dtype=np.float32, variant=""):
        assert variant == "" or variant in VARIANTS, variant
        self.variant = variant
        self.lib = C.CDLL(build(variant=variant))
        self.dtype = np.dtype(dtype)
        self.sfx = "_f32" if self.dtype == np.float32 else "_f64"
        self.real = C.c_float if self.dtype == np.float32 else C.c_double
        self.Params = _params_type(self.real)

    def _f(self, name):
        return getattr(self.lib, name + self.sfx)

    def _a(self, x, shape_last=3):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        return x

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def params(self, use_kdtree=True, center_reference=True, **kw):
        d = dict(DEFAULT_CHAIN)
        d.update(kw)
        # pair_order: the permutation of the reading's points the reduction tree takes the pairs in (icp_oracle.c "RT-1");
        # None = scan order.  E.g. the HIP path's sorting order of the same reading (icp.Context.reading_order).
        order = d.get("pair_order")
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int32)
        prm = self.Params(d["max_dist"], d["trim_ratio"], d["max_iters"], d["min_diff_rot"],
                          d["min_diff_trans"], d["smooth_length"], d["sensor_std_dev"],
                          int(use_kdtree), int(center_reference), d.get("outlier_max_dist", 0.0), d.get("quantile_scale", 1.0),
                          int(d.get("knn", 1)), int(d.get("error_minimizer", 0)), float(d.get("bound_max_rot", 0.0)),
                          float(d.get("bound_max_trans", 0.0)), float(d.get("normal_max_angle", 0.0)), int(d.get("robust_fct", 0)),
                          float(d.get("robust_tuning", 1.0)), int(d.get("robust_scale", 1)), float(d.get("robust_approx", 0.0)),
                          order.ctypes.data if order is not None else None)
        prm._keep_order = order                       # (the structure only holds the address)
        return prm

    # -- stages -----------------------------------------------------------
    def transform(self, T, pts, rotate_only=False):
        pts = self._a(pts)
        out = np.empty_like(pts)
        T = np.ascontiguousarray(T, dtype=np.float64)
        self._f("orc_transform")(self._p(T), self._p(pts), self._p(out), C.c_int(pts.shape[0]), C.c_int(int(rotate_only)))
        return out

    def centroid(self, pts):
        pts = self._a(pts)
        mean = np.zeros(3, dtype=self.dtype)
        self._f("orc_centroid")(self._p(pts), C.c_int(pts.shape[0]), self._p(mean))
        return mean

    def knn_brute(self, q, m, max_dist=np.inf):
        q, m = self._a(q), self._a(m)
        ids = np.empty(q.shape[0], dtype=np.int32)
        d2 = np.empty(q.shape[0], dtype=self.dtype)
        self._f("orc_knn_brute")(self._p(q), C.c_int(q.shape[0]), self._p(m), C.c_int(m.shape[0]),
                                 self.real(max_dist), self._p(ids), self._p(d2))
        return ids, d2

    def knn_kdtree(self, q, m, max_dist=np.inf):
        q, m = self._a(q), self._a(m)
        ids = np.empty(q.shape[0], dtype=np.int32)
        d2 = np.empty(q.shape[0], dtype=self.dtype)
        b = self._f("orc_kdtree_build"); b.restype = C.c_void_p
        t = b(self._p(m), C.c_int(m.shape[0]))
        self._f("orc_kdtree_knn")(C.c_void_p(t), self._p(q), C.c_int(q.shape[0]), self.real(max_dist),
                                  self._p(ids), self._p(d2))
        self._f("orc_kdtree_free")(C.c_void_p(t))
        return ids, d2

    def knn_brute_k(self, q, m, k, max_dist=np.inf):
        """brute-force k nearest neighbours: (ids (n,k), d2 (n,k)) in (d2, index) order, -1 / +inf where fewer lie within maxDist"""
        q, m = self._a(q), self._a(m)
        ids = np.empty((q.shape[0], k), dtype=np.int32)
        d2 = np.empty((q.shape[0], k), dtype=self.dtype)
        self._f("orc_knn_brute_k")(self._p(q), C.c_int(q.shape[0]), self._p(m), C.c_int(m.shape[0]), C.c_int(k),
                                   self.real(max_dist), self._p(ids), self._p(d2))
        return ids, d2

    def filter_chain(self, filters, features):
        """the input filters that only drop points (icp_oracle.c: orc_filter_chain): filters = [(type, p0, p1, ...)] with the type
        numbers of include/pgicp.h, features (n, frows); returns the kept input indices"""
        f = self._a(features)
        types = np.array([int(s[0]) for s in filters], dtype=np.int32)
        params = np.zeros((max(1, len(filters)), 8), dtype=np.float64)
        for k, s in enumerate(filters):
            params[k, :len(s) - 1] = s[1:]
        idx = np.empty(max(1, f.shape[0]), dtype=np.int32)
        n_out = C.c_int(0)
        st = self._f("orc_filter_chain")(C.c_int(len(filters)), self._p(types), self._p(params), self._p(f), C.c_int(f.shape[1]),
                                         C.c_int(f.shape[0]), self._p(idx), C.byref(n_out))
        assert st == 0, st
        return idx[:n_out.value].copy()

    def fixstep_next(self, step, start_step, end_step, step_mult):
        f = self.lib.orc_fixstep_next
        f.restype = C.c_double
        return f(C.c_double(step), C.c_double(start_step), C.c_double(end_step), C.c_double(step_mult))

    def shadow_keep(self, xyz, nrm, eps_angle=0.1):
        """[EXT] ShadowDataPointsFilter{eps} (orc_shadow_keep): boolean keep mask"""
        xyz, nrm = self._a(xyz), self._a(nrm)
        keep = np.zeros(len(xyz), dtype=np.int32)
        self._f("orc_shadow_keep")(self._p(xyz), self._p(nrm), C.c_int(len(xyz)), self.real(eps_angle), keep.ctypes.data_as(C.c_void_p))
        return keep.astype(bool)

    def sampling_surface_normal(self, xyz, knn=7, ratio=0.5, sampling_method=0, max_box_dim=np.inf, seed=1):
        """[EXT] SamplingSurfaceNormalDataPointsFilter (orc_sampling_surface_normal): dict(keep (n,) bool, normals (n,3) -- rows of kept
        points --, xyz (n,3): the kept points' coordinates (box means with sampling_method 1), boxes: boxes fused)"""
        xyz = self._a(xyz)
        n = len(xyz)
        keep = np.zeros(n, dtype=np.int32)
        nrm = np.zeros((n, 3), dtype=self.dtype)
        out = np.zeros((n, 3), dtype=self.dtype)
        f = self._f("orc_sampling_surface_normal")
        f.restype = C.c_int
        boxes = f(self._p(xyz), C.c_int(n), C.c_int(knn), self.real(ratio), C.c_int(sampling_method), self.real(max_box_dim), C.c_double(seed),
                  keep.ctypes.data_as(C.c_void_p), self._p(nrm), self._p(out))
        assert boxes >= 0
        return dict(keep=keep.astype(bool), normals=nrm, xyz=out, boxes=boxes)

    def densities(self, xyz, ids):
        """[EXT] SurfaceNormalDataPointsFilter{keepDensities}: neighbours / volume of the sphere that holds them around their mean"""
        xyz = self._a(xyz)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        d = np.empty(len(xyz), dtype=self.dtype)
        self._f("orc_densities")(self._p(xyz), C.c_int(len(xyz)), C.c_int(ids.shape[1]), self._p(ids), self._p(d))
        return d

    def max_density_keep(self, dens, max_density=10.0, seed=1):
        """[EXT] MaxDensityDataPointsFilter{maxDensity}: boolean keep mask"""
        dens = np.ascontiguousarray(dens, dtype=self.dtype)
        keep = np.zeros(len(dens), dtype=np.int32)
        self._f("orc_max_density_keep")(self._p(dens), C.c_int(len(dens)), self.real(max_density), C.c_double(seed), keep.ctypes.data_as(C.c_void_p))
        return keep.astype(bool)

    def simple_sensor_noise(self, xyz, sensor_type=0, gain=1.0):
        """[EXT] SimpleSensorNoiseDataPointsFilter{sensorType, gain}: the `simpleSensorNoise` descriptor (one value per point)"""
        xyz = self._a(xyz)
        out = np.empty(len(xyz), dtype=self.dtype)
        rc = self._f("orc_simple_sensor_noise")(self._p(xyz), C.c_int(len(xyz)), C.c_int(sensor_type), self.real(gain), self._p(out))
        if rc != 0:
            raise ValueError("SimpleSensorNoiseDataPointsFilter: sensorType must be 0 ... 4")
        return out

    def sensor_noise_overlap(self, d2, weights, noise):
        """[EXT] getOverlap() of a reading that carries `simpleSensorNoise`: d2 / weights (n,) or (n, knn) of the last iteration"""
        d2 = np.ascontiguousarray(d2, dtype=self.dtype)
        w = np.ascontiguousarray(weights, dtype=self.dtype)
        noise = np.ascontiguousarray(noise, dtype=self.dtype)
        knn = 1 if d2.ndim == 1 else d2.shape[1]
        f = self._f("orc_sensor_noise_overlap")
        f.restype = C.c_double
        return float(f(self._p(d2), self._p(w), self._p(noise), C.c_int(len(noise)), C.c_int(knn)))

    def robust_weights(self, d2, fct, tuning=1.0, scale=1, approx=0.0):
        """[EXT] RobustOutlierFilter (orc_robust_weights): (weights, squared scale)"""
        d2 = np.ascontiguousarray(d2, dtype=self.dtype)
        w = np.empty_like(d2)
        s2 = self.real(0)
        st = self._f("orc_robust_weights")(self._p(d2), C.c_int(d2.size), C.c_int(fct), self.real(tuning), C.c_int(scale), self.real(approx), self._p(w), C.byref(s2))
        assert st == 0, st
        return w, np.dtype(self.dtype).type(s2.value)

    def normal_weights(self, rd_nrm, ref_nrm, ids, max_angle, w=None):
        """[EXT] SurfaceNormalOutlierFilter{maxAngle}: multiplies its weights into w (ones by default); ids (n,) or (n,k)"""
        rd_nrm, ref_nrm = self._a(rd_nrm), self._a(ref_nrm)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        k = 1 if ids.ndim == 1 else ids.shape[1]
        w = np.ones(ids.shape, dtype=self.dtype) if w is None else np.ascontiguousarray(w, dtype=self.dtype).copy()
        self._f("orc_normal_weights")(self._p(rd_nrm), self._p(ref_nrm), self._p(ids), C.c_int(rd_nrm.shape[0]), C.c_int(k),
                                      self.real(max_angle), self._p(w))
        return w

    def p2point_system(self, p, ref_xyz, ids, w, order=None):
        p, ref_xyz = self._a(p), self._a(ref_xyz)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        w = np.ascontiguousarray(w, dtype=self.dtype)
        k = 1 if ids.ndim == 1 else ids.shape[1]
        sys_ = np.zeros(30, dtype=np.float64)
        keep, po = self._order(order)
        st = self._f("orc_p2point_system_o")(self._p(p), C.c_int(p.shape[0]), C.c_int(k), self._p(ref_xyz), self._p(ids), self._p(w),
                                             po, self._p(sys_))
        return st, sys_

    def solve_p2point(self, sys_):
        sys_ = np.ascontiguousarray(sys_, dtype=np.float64)
        T = np.zeros((4, 4))
        rank = self._f("orc_solve_p2point")(self._p(sys_), self._p(T))
        return T, rank

    def trim_weights(self, d2, ratio):
        d2 = np.ascontiguousarray(d2, dtype=self.dtype)
        w = np.empty_like(d2)
        limit = self.real(0)
        nf = C.c_int(0)
        st = self._f("orc_trim_weights")(self._p(d2), C.c_int(d2.shape[0]), self.real(ratio), self._p(w),
                                         C.byref(limit), C.byref(nf))
        return st, w, limit.value, nf.value

    def median_weights(self, d2, factor):
        """[EXT] MedianDistOutlierFilter{factor} -> (weights, limit, n_finite)"""
        d2 = np.ascontiguousarray(d2, dtype=self.dtype)
        w = np.empty_like(d2)
        limit = self.real(0)
        nf = C.c_int(0)
        st = self._f("orc_median_weights")(self._p(d2), C.c_int(d2.shape[0]), self.real(factor), self._p(w), C.byref(limit), C.byref(nf))
        assert st == 0
        return w, np.dtype(self.dtype).type(limit.value), nf.value

    @staticmethod
    def _order(order):
        """the reduction tree's order (icp_oracle.c "RT-1"): None = scan order, or a permutation of the points"""
        if order is None:
            return None, None
        o = np.ascontiguousarray(order, dtype=np.int32)
        return o, o.ctypes.data_as(C.c_void_p)

    def p2plane_system(self, p, ref_xyz, ref_nrm, ids, w, order=None):
        p, ref_xyz, ref_nrm = self._a(p), self._a(ref_xyz), self._a(ref_nrm)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        w = np.ascontiguousarray(w, dtype=self.dtype)
        k = 1 if ids.ndim == 1 else ids.shape[1]
        sys_ = np.zeros(30, dtype=np.float64)
        keep, po = self._order(order)
        st = self._f("orc_p2plane_system_o")(self._p(p), C.c_int(p.shape[0]), C.c_int(k), self._p(ref_xyz), self._p(ref_nrm),
                                             self._p(ids), self._p(w), po, self._p(sys_))
        return st, sys_

    def solve6(self, sys_):
        sys_ = np.ascontiguousarray(sys_, dtype=np.float64)
        x = np.zeros(6)
        rank = C.c_int(0)
        self._f("orc_solve6")(self._p(sys_), self._p(x), C.byref(rank))
        return x, rank.value

    def delta_T(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        T = np.zeros((4, 4))
        self._f("orc_delta_T")(self._p(x), self._p(T))
        return T

    def covariance(self, p, ref_xyz, ref_nrm, ids, w, dT, sensor_std_dev, order=None):
        p, ref_xyz, ref_nrm = self._a(p), self._a(ref_xyz), self._a(ref_nrm)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        w = np.ascontiguousarray(w, dtype=self.dtype)
        dT = np.ascontiguousarray(dT, dtype=np.float64)
        k = 1 if ids.ndim == 1 else ids.shape[1]
        cov = np.zeros((6, 6))
        keep, po = self._order(order)
        self._f("orc_covariance_o")(self._p(p), C.c_int(p.shape[0]), C.c_int(k), self._p(ref_xyz), self._p(ref_nrm), self._p(ids),
                                    self._p(w), self._p(dT), C.c_double(sensor_std_dev), po, self._p(cov))
        return cov

    def knn_k(self, ref, q, k, max_dist=np.inf):
        """k nearest neighbours of every q in ref: (ids (n,k), d2 (n,k)), lexicographic (d2, index) order."""
        ref, q = self._a(ref), self._a(q)
        f = self._f("orc_kdtree_build"); f.restype = C.c_void_p
        t = C.c_void_p(f(self._p(ref), C.c_int(ref.shape[0])))
        ids = np.empty((q.shape[0], k), dtype=np.int32)
        d2 = np.empty((q.shape[0], k), dtype=self.dtype)
        self._f("orc_kdtree_knn_k")(t, self._p(q), C.c_int(q.shape[0]), C.c_int(k), self.real(max_dist), self._p(ids), self._p(d2))
        self._f("orc_kdtree_free")(t)
        return ids, d2

    def surface_normals(self, xyz, knn, max_dist=np.inf):
        """SurfaceNormalDataPointsFilter: dict(normals (n,3), eigen_values (n,3) ascending, ids (n,knn), d2 (n,knn))."""
        xyz = self._a(xyz)
        n = xyz.shape[0]
        nrm = np.empty((n, 3), dtype=self.dtype)
        ev = np.empty((n, 3), dtype=self.dtype)
        ids = np.empty((n, knn), dtype=np.int32)
        d2 = np.empty((n, knn), dtype=self.dtype)
        st = self._f("orc_surface_normals")(self._p(xyz), C.c_int(n), C.c_int(knn), self.real(max_dist), self._p(nrm), self._p(ev),
                                            self._p(ids), self._p(d2))
        assert st == 0
        return dict(normals=nrm, eigen_values=ev, ids=ids, d2=d2)

    def build_local_map(self, clouds_xyz, clouds_nrm, T_ref_kf):
        """clouds[0] is the reference keyframe; T_ref_kf[k] moves cloud k into its frame."""
        k = len(clouds_xyz)
        xs = [self._a(c) for c in clouds_xyz]
        ns = [self._a(c) for c in clouds_nrm]
        counts = np.array([c.shape[0] for c in xs], dtype=np.int32)
        Ts = np.ascontiguousarray(np.stack(T_ref_kf), dtype=np.float64)
        out_x = np.empty((int(counts.sum()), 3), dtype=self.dtype)
        out_n = np.empty_like(out_x)
        PP = C.c_void_p * k
        self._f("orc_build_local_map")(C.c_int(k), PP(*[c.ctypes.data for c in xs]), PP(*[c.ctypes.data for c in ns]),
                                       self._p(counts), self._p(Ts), self._p(out_x), self._p(out_n))
        return out_x, out_n

    def partial_chain(self, reading, ref_xyz, ref_nrm, T, **kw):
        """Localizer::ComputeOverlapWith / LoopCloser::ComputeResidualError."""
        reading, ref_xyz, ref_nrm = self._a(reading), self._a(ref_xyz), self._a(ref_nrm)
        prm = self.params(**kw)
        T = np.ascontiguousarray(T, dtype=np.float64)
        ov, rs = C.c_double(0), C.c_double(0)
        k = max(1, int(prm.knn))
        shape = (reading.shape[0],) if k == 1 else (reading.shape[0], k)
        ids = np.empty(shape, dtype=np.int32)
        d2 = np.empty(shape, dtype=self.dtype)
        st = self._f("orc_partial_chain")(C.byref(prm), self._p(reading), C.c_int(reading.shape[0]), self._p(ref_xyz),
                                          self._p(ref_nrm), C.c_int(ref_xyz.shape[0]), self._p(T), C.byref(ov),
                                          C.byref(rs), self._p(ids), self._p(d2))
        return dict(status=st, overlap=ov.value, residual=rs.value, ids=ids, d2=d2)

    def icp(self, reading, ref_xyz, ref_nrm, T_init, trace=False, reading_nrm=None, want_last=True, **kw):
        """reading_nrm: the reading's `normals` descriptor (only the SurfaceNormalOutlierFilter looks at it); with knn > 1
        last_ids / last_d2 are (n, knn)"""
        reading, ref_xyz, ref_nrm = self._a(reading), self._a(ref_xyz), self._a(ref_nrm)
        rn = self._a(reading_nrm) if reading_nrm is not None else None
        prm = self.params(**kw)
        T_init = np.ascontiguousarray(T_init, dtype=np.float64)
        T_out = np.zeros((4, 4))
        res = Result()
        cap = prm.max_iters if trace else 0
        tr = np.zeros((max(cap, 1), 4, 4))
        k = max(1, int(prm.knn))
        shape = (reading.shape[0],) if k == 1 else (reading.shape[0], k)
        ids = np.empty(shape, dtype=np.int32)
        d2 = np.empty(shape, dtype=self.dtype)
        st = self._f("orc_icp_ex")(C.byref(prm), self._p(reading), self._p(rn) if rn is not None else None, C.c_int(reading.shape[0]),
                                   self._p(ref_xyz), self._p(ref_nrm), C.c_int(ref_xyz.shape[0]), self._p(T_init), self._p(T_out),
                                   C.byref(res), self._p(tr) if trace else None, C.c_int(cap), self._p(ids), self._p(d2))
        out = dict(status=st, T=T_out, iterations=res.iterations, converged=bool(res.converged),
                   max_iter_reached=bool(res.max_iter_reached), overlap=res.overlap, residual=res.residual,
                   trim_limit=res.trim_limit, n_kept=res.n_kept, n_finite=res.n_finite,
                   cov=np.array(res.cov[:]).reshape(6, 6), last_ids=ids, last_d2=d2)
        if trace:
            out["trace"] = tr[: res.iterations]
        return out

    # -- prebuilt reference (setMap amortisation) ---------------------------
    def map_create(self, ref_xyz, ref_nrm, center=True, use_kdtree=True):
        ref_xyz, ref_nrm = self._a(ref_xyz), self._a(ref_nrm)
        f = self._f("orc_map_create"); f.restype = C.c_void_p
        h = f(self._p(ref_xyz), self._p(ref_nrm), C.c_int(ref_xyz.shape[0]), C.c_int(int(center)), C.c_int(int(use_kdtree)))
        return (C.c_void_p(h), ref_xyz, ref_nrm)            # keep the borrowed arrays alive

    def map_free(self, m):
        self._f("orc_map_free")(m[0])

    def icp_map(self, m, reading, T_init, want_last=False, **kw):
        """want_last: also return the last iteration's correspondences (last_ids, last_d2)"""
        reading = self._a(reading)
        prm = self.params(**kw)
        T_init = np.ascontiguousarray(T_init, dtype=np.float64)
        T_out = np.zeros((4, 4))
        res = Result()
        k = max(1, int(prm.knn))
        shape = (reading.shape[0],) if k == 1 else (reading.shape[0], k)
        ids = np.empty(shape, dtype=np.int32) if want_last else None
        d2 = np.empty(shape, dtype=self.dtype) if want_last else None
        st = self._f("orc_icp_map")(C.byref(prm), m[0], self._p(reading), C.c_int(reading.shape[0]), self._p(T_init),
                                    self._p(T_out), C.byref(res), None, C.c_int(0),
                                    self._p(ids) if want_last else None, self._p(d2) if want_last else None)
        out = dict(status=st, T=T_out, iterations=res.iterations, converged=bool(res.converged),
                   max_iter_reached=bool(res.max_iter_reached), overlap=res.overlap, residual=res.residual,
                   trim_limit=res.trim_limit, n_kept=res.n_kept, n_finite=res.n_finite, cov=np.array(res.cov[:]).reshape(6, 6))
        if want_last:
            out.update(last_ids=ids, last_d2=d2)
        return out

    # -- checker (type independent) ---------------------------------------
    def checker(self, max_iters, min_rot, min_trans, smooth):
        c = Checker()
        self.lib.orc_checker_init(C.byref(c), C.c_int(max_iters), C.c_double(min_rot), C.c_double(min_trans), C.c_int(smooth))
        return c

    def checker_set_bound(self, c, max_rot, max_trans):
        self.lib.orc_checker_set_bound(C.byref(c), C.c_double(max_rot), C.c_double(max_trans))

    def checker_check(self, c, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        return self.lib.orc_checker_check(C.byref(c), self._p(T))
