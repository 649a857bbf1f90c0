"""Python binding of the C ABI in include/pgicp.h (ctypes, no torch types cross it).

This is plumbing for tests and bench.py; the C++ drop-in host layer lives in
include/pgslam_amd/.  The names mirror the libpointmatcher objects pgslam drives
(reference src/pgslam/Localizer.hpp:126,148,309-347, LoopCloser.hpp:98,343-365):

    ICPSequence.setMap(cloud)            -> Context.set_map(xyz, normals)
    ICPSequence.__call__(reading, T)     -> Context.align(map, reading, T_init)
    ICP.__call__(reading, reference, T)  -> Context.icp_pair(...)
    matcher.findClosests                 -> Context.match(...)
    outlierFilters.compute               -> Context.outlier_weights(...)
    ErrorElements / getResidualError     -> Context.error_stats(...)

There is no CPU fallback: if libpgicp.so is missing, or no GPU is present,
construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGICP_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libpgicp.so")

OK, ERR_NO_MATCH, ERR_NAN, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_NOT_RIGID, ERR_BOUND = range(8)
MINIMIZER_POINT_TO_PLANE, MINIMIZER_POINT_TO_POINT, MINIMIZER_POINT_TO_PLANE_4DOF, MINIMIZER_POINT_TO_POINT_WITH_COV = 0, 1, 2, 3
HOST, DEVICE, HOST_PINNED = 0, 1, 2
MATCHER_GRID, MATCHER_BRUTE = 0, 1
PROF_NAMES = ["knn_grid", "knn_brute", "trim_select", "p2plane_reduce", "solve_update", "pretransform",
              "covariance", "grid_build", "knn_slow", "surface_normals", "knn_grid_unseeded"]

# every symbol include/pgicp.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "pgicp_abi_version", "pgicp_device_count", "pgicp_ctx_create", "pgicp_ctx_destroy", "pgicp_last_error",
    "pgicp_ctx_stream", "pgicp_ctx_synchronize", "pgicp_default_params", "pgicp_set_params", "pgicp_get_params",
    "pgicp_map_create_f32", "pgicp_map_create_f64", "pgicp_map_create_batch_f32", "pgicp_map_create_batch_f64",
    "pgicp_map_destroy", "pgicp_map_size", "pgicp_map_transfer",
    "pgicp_align_f32", "pgicp_align_f64", "pgicp_align_batch_f32", "pgicp_align_batch_f64",
    "pgicp_align_residual_batch_f32", "pgicp_align_residual_batch_f64", "pgicp_filter_cloud_f32", "pgicp_filter_cloud_f64",
    "pgicp_device_alloc", "pgicp_device_free", "pgicp_device_copy",
    "pgicp_icp_pair_f32", "pgicp_icp_pair_f64", "pgicp_match_f32", "pgicp_match_f64",
    "pgicp_outlier_weights_f32", "pgicp_outlier_weights_f64", "pgicp_error_stats_f32", "pgicp_error_stats_f64",
    "pgicp_partial_chain_f32", "pgicp_partial_chain_f64", "pgicp_partial_chain_batch_f32",
    "pgicp_partial_chain_batch_f64", "pgicp_transform_f32", "pgicp_transform_f64",
    "pgicp_build_local_map_f32", "pgicp_build_local_map_f64", "pgicp_surface_normals_f32", "pgicp_surface_normals_f64", "pgicp_shard_pairs", "pgicp_check_icp_result",
    "pgicp_profile_enable", "pgicp_profile_reset", "pgicp_profile_get", "pgicp_debug_counters", "pgicp_debug_alloc_stats", "pgicp_ctx_create_priority", "pgicp_filter_cloud_dev_f32", "pgicp_filter_cloud_dev_f64",
    "pgicp_debug_last_matches_f32", "pgicp_debug_last_matches_f64",
    "pgicp_status_string", "pgicp_upload_f32", "pgicp_upload_f64", "pgicp_host_alloc", "pgicp_host_free",
    "pgicp_ctx_device", "pgicp_comm_unique_id", "pgicp_comm_create", "pgicp_comm_destroy", "pgicp_comm_info",
    "pgicp_comm_last_error", "pgicp_shard_slots", "pgicp_allgather_edges", "pgicp_comm_create_host", "pgicp_profile_process",
    "pgicp_debug_reading_order", "pgicp_partial_chain_seeded_f32", "pgicp_partial_chain_seeded_f64",
]
SUM_ORDER_SORTED, SUM_ORDER_SCAN = 0, 1


class Params(C.Structure):
    _fields_ = [("knn", C.c_int), ("epsilon", C.c_double), ("max_dist", C.c_double), ("trim_ratio", C.c_double),
                ("max_iters", C.c_int), ("min_diff_rot", C.c_double), ("min_diff_trans", C.c_double),
                ("smooth_length", C.c_int), ("sensor_std_dev", C.c_double), ("matcher", C.c_int),
                ("grid_cell", C.c_double), ("check_every", C.c_int), ("outlier_max_dist", C.c_double),
                ("quantile_scale", C.c_double), ("error_minimizer", C.c_int), ("bound_max_rot", C.c_double),
                ("bound_max_trans", C.c_double), ("normal_max_angle", C.c_double), ("robust_fct", C.c_int), ("robust_tuning", C.c_double),
                ("robust_scale", C.c_int), ("robust_approx", C.c_double), ("sum_order", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("converged", C.c_int), ("max_iter_reached", C.c_int),
                ("overlap", C.c_double), ("residual", C.c_double), ("trim_limit", C.c_double), ("n_kept", C.c_int),
                ("n_finite", C.c_int), ("cov", C.c_double * 36)]

    def as_dict(self):
        return dict(status=self.status, iterations=self.iterations, converged=bool(self.converged),
                    max_iter_reached=bool(self.max_iter_reached), overlap=self.overlap, residual=self.residual,
                    trim_limit=self.trim_limit, n_kept=self.n_kept, n_finite=self.n_finite,
                    cov=np.array(self.cov[:]).reshape(6, 6))


class Problem(C.Structure):
    _fields_ = [("map_id", C.c_int), ("reading", C.c_void_p), ("stride", C.c_int), ("n", C.c_int), ("mem", C.c_int),
                ("T_init", C.c_double * 16), ("normals", C.c_void_p), ("nstride", C.c_int)]


class Filter(C.Structure):
    _fields_ = [("type", C.c_int), ("p", C.c_double * 8)]


FILTER_IDENTITY, FILTER_MAX_DIST, FILTER_MIN_DIST, FILTER_BOUNDING_BOX, FILTER_REMOVE_NAN, FILTER_FIX_STEP, FILTER_RANDOM_SAMPLING, \
    FILTER_MAX_POINT_COUNT = range(8)


class Edge(C.Structure):
    _fields_ = [("from_id", C.c_int64), ("to_id", C.c_int64), ("accepted", C.c_int32), ("status", C.c_int32),
                ("iterations", C.c_int32), ("max_iter_reached", C.c_int32), ("overlap", C.c_double),
                ("residual", C.c_double), ("T_from_to", C.c_double * 16), ("cov", C.c_double * 36),
                ("reserved", C.c_double * 6)]


assert C.sizeof(Edge) == 512


class PgicpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pgicp error {code}: {msg}")
        self.code = code


class ConvergenceError(PgicpError):
    """PM::ConvergenceError: 'no point to minimize' / NaN in the checkers / BoundTransformationChecker's limit exceeded."""


_lib = None


def load_library() -> C.CDLL:
    """dlopen libpgicp.so.  Raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                              "pgslam_amd has no CPU fallback")
        # PyTorch wheels carry their own libamdhip64.  If libpgicp (linked against /opt/rocm's copy) enters the process
        # first and torch later, the process holds two HIP runtimes and torch reports "No HIP GPUs are available"; with
        # torch loaded first the loader resolves libpgicp's dependency to the copy that is already there.
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.pgicp_last_error.restype = C.c_char_p
        _lib.pgicp_status_string.restype = C.c_char_p
        _lib.pgicp_comm_last_error.restype = C.c_char_p
        _lib.pgicp_ctx_stream.restype = C.c_void_p
    return _lib


_TORCH_DTYPES = {}          # torch dtype -> numpy dtype (the string round trip costs microseconds per reading of a batch)

# numpy views of the two ctypes records a batch call exchanges (same layout: ctypes' natural alignment)
_PROBLEM_DTYPE = np.dtype({"names": ["map_id", "reading", "stride", "n", "mem", "T_init", "normals", "nstride"],
                           "formats": ["<i4", "<u8", "<i4", "<i4", "<i4", ("<f8", (16,)), "<u8", "<i4"],
                           "offsets": [0, 8, 16, 20, 24, 32, 160, 168], "itemsize": 176})
assert C.sizeof(Problem) == _PROBLEM_DTYPE.itemsize
_STATS_DTYPE = np.dtype({"names": ["status", "iterations", "converged", "max_iter_reached", "overlap", "residual", "trim_limit",
                                   "n_kept", "n_finite", "cov"],
                         "formats": ["<i4", "<i4", "<i4", "<i4", "<f8", "<f8", "<f8", "<i4", "<i4", ("<f8", (36,))],
                         "offsets": [0, 4, 8, 12, 16, 24, 32, 40, 44, 48], "itemsize": 336})


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DevPtr:
    """A reading already on the device, as pgicp_upload_* hands it out: raw address, stride, point count."""

    def __init__(self, ptr, stride, n, dtype, keep=None):
        self.ptr, self.stride, self.n, self.dtype, self.keep = ptr, stride, n, np.dtype(dtype), keep


class _Buf:
    """(pointer, stride, n, mem, dtype) view of a numpy array or a torch CUDA tensor.
    Accepts (N,3) packed xyz or (N,4) homogeneous rows (= libpointmatcher's 4xN
    column-major `features`)."""

    def __init__(self, x, dtype=None):
        if isinstance(x, DevPtr):
            self.keep, self.ptr, self.stride, self.n, self.mem, self.dtype = x, x.ptr, x.stride, x.n, DEVICE, x.dtype
            return
        if _is_torch(x):
            if not x.is_cuda:
                x = x.numpy()
            else:
                assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] >= 3
                self.keep = x
                self.ptr = x.data_ptr()
                self.stride = x.stride(0)
                self.n = x.shape[0]
                self.mem = DEVICE
                dt = _TORCH_DTYPES.get(x.dtype)
                if dt is None:
                    dt = _TORCH_DTYPES[x.dtype] = np.dtype(str(x.dtype).replace("torch.", ""))
                self.dtype = dt
                return
        x = np.asarray(x)
        if dtype is not None:
            x = x.astype(dtype, copy=False)
        assert x.ndim == 2 and x.shape[1] >= 3 and x.strides[1] == x.itemsize
        self.keep = x
        self.ptr = x.ctypes.data
        self.stride = x.strides[0] // x.itemsize
        self.n = x.shape[0]
        self.mem = HOST
        self.dtype = x.dtype


def _bufs(xs, dtype=None):
    """_Buf views of a list of clouds; the same object listed twice (a keyframe that is the reference of several loop-closure
    candidates) is looked at once -- 1.2 us per torch tensor, three lists of 512 in a loop-closure step."""
    seen = {}
    out = []
    for x in xs:
        b = seen.get(id(x))
        if b is None:
            b = seen[id(x)] = _Buf(x, dtype)
        out.append(b)
    return out


def _T16(T):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4))
    return (C.c_double * 16)(*T.ravel())


UNIQUE_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """pgicp_comm_unique_id (rank 0); pass the bytes to the other ranks by any channel."""
    lib = load_library()
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    st = lib.pgicp_comm_unique_id(buf)
    if st != OK:
        raise PgicpError(st, lib.pgicp_comm_last_error().decode())
    return buf.raw


def shard_slots(costs, world_size: int) -> int:
    """Largest shard of the deterministic split: the block size of the edge all-gather, known to every rank."""
    lib = load_library()
    n = len(costs)
    arr = (C.c_int64 * max(n, 1))(*[int(c) for c in costs])
    out = C.c_int()
    st = lib.pgicp_shard_slots(C.c_int(n), arr, C.c_int(world_size), C.byref(out))
    if st != OK:
        raise PgicpError(st, "pgicp_shard_slots failed")
    return out.value


class Comm:
    """RCCL communicator of one rank (pgicp_comm_create) on a context's device."""

    def __init__(self, ctx: "Context", world_size: int, rank: int, unique_id: bytes):
        self.lib = ctx.lib
        self.ctx = ctx
        self.world_size, self.rank = world_size, rank
        h = C.c_void_p()
        st = self.lib.pgicp_comm_create(ctx.h, C.c_int(world_size), C.c_int(rank), C.create_string_buffer(unique_id, UNIQUE_ID_BYTES), C.byref(h))
        if st != OK:
            raise PgicpError(st, self.lib.pgicp_comm_last_error().decode())
        self.h = h

    @classmethod
    def host(cls, world_size: int, rank: int, shm_path: str, max_slots_per_rank: int) -> "Comm":
        """pgicp_comm_create_host: the same collective with the blocks travelling through a shared-memory file
        (no device; the host 'fake all-gather' of SURVEY.md section 4 T4)."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.ctx = None
        self.world_size, self.rank = world_size, rank
        h = C.c_void_p()
        st = self.lib.pgicp_comm_create_host(C.c_int(world_size), C.c_int(rank), shm_path.encode(), C.c_int(max_slots_per_rank), C.byref(h))
        if st != OK:
            raise PgicpError(st, self.lib.pgicp_comm_last_error().decode())
        self.h = h
        return self

    def info(self):
        """(world_size, rank) as the communicator itself reports them (pgicp_comm_info)."""
        w, r = C.c_int(), C.c_int()
        self.lib.pgicp_comm_info(self.h, C.byref(w), C.byref(r))
        return w.value, r.value

    def allgather_edges(self, local_edges: np.ndarray, pair_index, slots_per_rank: int, n_total: int) -> np.ndarray:
        """pgicp_allgather_edges on numpy records of the 512-byte pgicp_edge layout."""
        assert local_edges.dtype.itemsize == C.sizeof(Edge)
        local = np.ascontiguousarray(local_edges)
        idx = np.ascontiguousarray(np.asarray(pair_index, dtype=np.int32))
        out = np.zeros(n_total, dtype=local_edges.dtype)
        st = self.lib.pgicp_allgather_edges(self.h, C.c_void_p(local.ctypes.data), C.c_void_p(idx.ctypes.data), C.c_int(len(local)),
                                            C.c_int(slots_per_rank), C.c_int(n_total), C.c_void_p(out.ctypes.data))
        if st != OK:
            raise PgicpError(st, self.lib.pgicp_comm_last_error().decode())
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.pgicp_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    def __init__(self, device: int = 0, **params):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.pgicp_ctx_create(C.c_int(device), C.byref(h))
        if st != OK:
            raise PgicpError(st, "pgicp_ctx_create failed" + (": PGICP_ERR_NO_DEVICE (no usable gfx950 device for this rank; the product has no CPU path)" if st == ERR_NO_DEVICE else ""))
        self.h = h
        self.device = device
        self.params = Params()
        self.lib.pgicp_default_params(C.byref(self.params))
        if params:
            self.set_params(**params)

    def close(self):
        if getattr(self, "h", None):
            self.lib.pgicp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st == OK:
            return
        msg = self.lib.pgicp_last_error(self.h).decode()
        if st in (ERR_NO_MATCH, ERR_NAN, ERR_BOUND):
            raise ConvergenceError(st, msg)
        raise PgicpError(st, msg)

    # ---- host-input pipeline ------------------------------------------------
    def host_alloc(self, shape, dtype=np.float32):
        """numpy array in pinned host memory (pgicp_host_alloc); release it with host_free(array)."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self.lib.pgicp_host_alloc(self.h, C.c_size_t(nbytes), C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        p = self._pinned.pop(a.ctypes.data)
        self._check(self.lib.pgicp_host_free(self.h, C.c_void_p(p)))

    # ---- device memory the caller owns between calls (keyframe clouds of a local map stay in HBM) ----
    def device_empty(self, n, stride=3, dtype=np.float32):
        """pgicp_device_alloc: room for n points at `stride` elements each; a DevPtr (release it with device_free)."""
        dtype = np.dtype(dtype)
        p = C.c_void_p()
        self._check(self.lib.pgicp_device_alloc(self.h, C.c_size_t(max(1, n * stride) * dtype.itemsize), C.byref(p)))
        return DevPtr(p.value, stride, n, dtype)

    def device_cloud(self, xyz, dtype=None):
        """A host cloud copied to device memory of its own (pgicp_device_alloc + pgicp_device_copy); strides as on the host."""
        b = _Buf(xyz, dtype)
        assert b.mem == HOST
        d = self.device_empty(b.n, b.stride, b.dtype)
        if b.n:
            nbytes = ((b.n - 1) * b.stride + 3) * b.dtype.itemsize
            self._check(self.lib.pgicp_device_copy(self.h, C.c_void_p(d.ptr), C.c_void_p(b.ptr), C.c_size_t(nbytes), C.c_int(0)))
        return d

    def device_download(self, d):
        """numpy (n, 3) copy of a DevPtr cloud"""
        out = np.zeros((d.n, d.stride), dtype=d.dtype)
        if d.n:
            nbytes = ((d.n - 1) * d.stride + 3) * d.dtype.itemsize
            self._check(self.lib.pgicp_device_copy(self.h, C.c_void_p(out.ctypes.data), C.c_void_p(d.ptr), C.c_size_t(nbytes), C.c_int(1)))
        return out[:, :3]

    def device_free(self, d):
        self._check(self.lib.pgicp_device_free(self.h, C.c_void_p(d.ptr)))
        d.ptr = None

    def upload(self, readings, pinned=False, dtype=None):
        """pgicp_upload_*: start the H2D transfer of host readings on the copy stream, return DevPtr handles at once."""
        bufs = [_Buf(r, dtype) for r in readings]
        assert all(b.mem == HOST for b in bufs)
        n = len(bufs)
        ct = C.c_float if bufs[0].dtype == np.float32 else C.c_double
        hosts = (C.c_void_p * n)(*[b.ptr for b in bufs])
        strides = (C.c_int * n)(*[b.stride for b in bufs])
        counts = (C.c_int * n)(*[b.n for b in bufs])
        out = (C.c_void_p * n)()
        fn = getattr(self.lib, "pgicp_upload" + self._sfx(bufs[0].dtype))
        self._check(fn(self.h, C.c_int(n), hosts, strides, counts, C.c_int(HOST_PINNED if pinned else HOST), out))
        del ct
        return [DevPtr(out[k], bufs[k].stride, bufs[k].n, bufs[k].dtype, keep=bufs[k].keep) for k in range(n)]

    def set_params(self, **kw):
        before = Params.from_buffer_copy(self.params)
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise KeyError(k)
            setattr(self.params, k, v)
        try:
            self._check(self.lib.pgicp_set_params(self.h, C.byref(self.params)))
        except PgicpError:
            self.params = before            # a refused setting leaves the context (and this mirror of it) as it was
            raise

    @property
    def stream(self):
        return self.lib.pgicp_ctx_stream(self.h)

    def synchronize(self):
        self._check(self.lib.pgicp_ctx_synchronize(self.h))

    @staticmethod
    def _sfx(dtype):
        return "_f32" if np.dtype(dtype) == np.float32 else "_f64"

    # ---- map -------------------------------------------------------------
    def set_map(self, xyz, normals=None, center=True, dtype=None) -> int:
        x = _Buf(xyz, dtype)
        nb = _Buf(normals, x.dtype) if normals is not None else None
        assert nb is None or (nb.n == x.n and nb.mem == x.mem)
        mid = C.c_int(-1)
        fn = getattr(self.lib, "pgicp_map_create" + self._sfx(x.dtype))
        self._check(fn(self.h, C.c_void_p(x.ptr), C.c_int(x.stride), C.c_void_p(nb.ptr if nb else None),
                       C.c_int(nb.stride if nb else 0), C.c_int(x.n), C.c_int(x.mem), C.c_int(int(center)), C.byref(mid)))
        return mid.value

    def set_maps(self, xyzs, normals=None, center=True, dtype=None):
        """Index several reference clouds with one host round trip (pgicp_map_create_batch)."""
        n = len(xyzs)
        xb = _bufs(xyzs, dtype)
        nb = _bufs(normals, xb[0].dtype) if normals is not None else None
        assert all(b.mem == xb[0].mem and b.dtype == xb[0].dtype for b in xb)
        assert nb is None or all(b.n == x.n and b.mem == x.mem for b, x in zip(nb, xb))
        ptrs = (C.c_void_p * n)(*[b.ptr for b in xb])
        strides = (C.c_int * n)(*[b.stride for b in xb])
        sizes = (C.c_int * n)(*[b.n for b in xb])
        nptrs = (C.c_void_p * n)(*[b.ptr for b in nb]) if nb else None
        nstrides = (C.c_int * n)(*[b.stride for b in nb]) if nb else None
        ids = (C.c_int * n)()
        fn = getattr(self.lib, "pgicp_map_create_batch" + self._sfx(xb[0].dtype))
        self._check(fn(self.h, C.c_int(n), ptrs, strides, nptrs, nstrides, sizes, C.c_int(xb[0].mem), C.c_int(int(center)), ids))
        return list(ids)

    def destroy_map(self, map_id):
        self._check(self.lib.pgicp_map_destroy(self.h, C.c_int(map_id)))

    def map_size(self, map_id):
        m = C.c_int(0)
        self._check(self.lib.pgicp_map_size(self.h, C.c_int(map_id), C.byref(m)))
        return m.value

    # ---- full ICP -----------------------------------------------------------
    def align(self, map_id, reading, T_init, dtype=None, normals=None):
        """normals: the reading's `normals` descriptor (a SurfaceNormalOutlierFilter in the chain needs it)"""
        if normals is not None:
            T, st = self.align_batch(map_id, [reading], [T_init], dtype=dtype, normals=[normals])
            return T[0], st[0]
        r = _Buf(reading, dtype)
        T_out = (C.c_double * 16)()
        st = Stats()
        fn = getattr(self.lib, "pgicp_align" + self._sfx(r.dtype))
        self._check(fn(self.h, C.c_int(map_id), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_int(r.mem),
                       _T16(T_init), T_out, C.byref(st)))
        return np.array(T_out[:]).reshape(4, 4), st.as_dict()

    def align_residual_batch(self, map_ids, readings, T_inits, dtype=None, normals=None, raw_stats=False):
        """pgicp_align_residual_batch: the ICPs of a batch of loop-closure candidates and, fused, the residual check of every
        result (LoopCloser.hpp:98, 343-365).  Returns (T (P,4,4), stats, residual (P,), ratio (P,), status (P,)); never raises
        for a failed candidate (its residual is +inf)."""
        return self.align_batch(map_ids, readings, T_inits, dtype=dtype, raise_on_error=False, normals=normals, _residual=True, _raw_stats=raw_stats)

    def align_batch(self, map_ids, readings, T_inits, dtype=None, raise_on_error=True, normals=None, _residual=False, _raw_stats=False):
        """`_raw_stats`: the pgicp_stats records come back as ONE numpy record array (fields as in include/pgicp.h) instead of a
        list of dicts -- a batch of 512 loop-closure candidates does not need 512 dictionaries to fill 512 edge records."""
        P = len(readings)
        if isinstance(map_ids, int):
            map_ids = [map_ids] * P
        bufs = _bufs(readings, dtype)
        nbufs = _bufs(normals, bufs[0].dtype) if normals is not None else None
        # the records are filled and read back through numpy views, column by column: per-problem attribute access on
        # ctypes structures cost ~9 us a problem, 1.2 ms of a 15 ms step at 128 problems
        pa = np.zeros(P, dtype=_PROBLEM_DTYPE)
        pa["map_id"] = map_ids
        pa["reading"] = [b.ptr for b in bufs]
        pa["stride"] = [b.stride for b in bufs]
        pa["n"] = [b.n for b in bufs]
        pa["mem"] = [b.mem for b in bufs]
        pa["T_init"] = np.asarray(T_inits, dtype=np.float64).reshape(P, 16)
        if nbufs is not None:
            assert all(nb.n == b.n and nb.mem == b.mem for nb, b in zip(nbufs, bufs))
            pa["normals"] = [b.ptr for b in nbufs]
            pa["nstride"] = [b.stride for b in nbufs]
        T_out = np.empty((P, 4, 4), dtype=np.float64)
        sa = np.zeros(P, dtype=_STATS_DTYPE)
        if _residual:
            res, ratio, rst = np.zeros(P), np.zeros(P), np.zeros(P, dtype=np.int32)
            fn = getattr(self.lib, "pgicp_align_residual_batch" + self._sfx(bufs[0].dtype))
            rc = fn(self.h, C.c_int(P), C.c_void_p(pa.ctypes.data), C.c_void_p(T_out.ctypes.data), C.c_void_p(sa.ctypes.data),
                    C.c_void_p(res.ctypes.data), C.c_void_p(ratio.ctypes.data), C.c_void_p(rst.ctypes.data))
        else:
            fn = getattr(self.lib, "pgicp_align_batch" + self._sfx(bufs[0].dtype))
            rc = fn(self.h, C.c_int(P), C.c_void_p(pa.ctypes.data), C.c_void_p(T_out.ctypes.data), C.c_void_p(sa.ctypes.data))
        if raise_on_error:
            self._check(rc)
        elif rc not in (OK, ERR_NO_MATCH, ERR_NAN, ERR_BOUND):
            self._check(rc)
        if _raw_stats:
            return (T_out, sa, res, ratio, rst) if _residual else (T_out, sa)
        cov = sa["cov"].reshape(P, 6, 6)
        cols = [sa[k].tolist() for k in ("status", "iterations", "converged", "max_iter_reached", "overlap", "residual", "trim_limit",
                                          "n_kept", "n_finite")]
        stats = [dict(status=st, iterations=it, converged=bool(cv), max_iter_reached=bool(mx), overlap=ov, residual=rs, trim_limit=tl,
                      n_kept=nk, n_finite=nf, cov=cov[p])
                 for p, (st, it, cv, mx, ov, rs, tl, nk, nf) in enumerate(zip(*cols))]
        if _residual:
            return T_out, stats, res, ratio, rst
        return T_out, stats

    def icp_pair(self, reading, ref_xyz, ref_nrm, T_init, dtype=None):
        r = _Buf(reading, dtype)
        x = _Buf(ref_xyz, r.dtype)
        nb = _Buf(ref_nrm, r.dtype)
        assert r.mem == x.mem == nb.mem
        T_out = (C.c_double * 16)()
        st = Stats()
        fn = getattr(self.lib, "pgicp_icp_pair" + self._sfx(r.dtype))
        self._check(fn(self.h, C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_void_p(x.ptr), C.c_int(x.stride),
                       C.c_void_p(nb.ptr), C.c_int(nb.stride), C.c_int(x.n), C.c_int(r.mem), _T16(T_init), T_out,
                       C.byref(st)))
        return np.array(T_out[:]).reshape(4, 4), st.as_dict()

    # ---- stages -----------------------------------------------------------
    def match(self, map_id, reading, T=None, dtype=None):
        r = _Buf(reading, dtype)
        fn = getattr(self.lib, "pgicp_match" + self._sfx(r.dtype))
        Tp = _T16(T) if T is not None else None
        k = max(1, int(self.params.knn))
        shape = (r.n,) if k == 1 else (r.n, k)              # knn entries per reading point
        if r.mem == DEVICE:
            import torch
            ids = torch.empty(shape, dtype=torch.int32, device=r.keep.device)
            d2 = torch.empty(shape, dtype=r.keep.dtype, device=r.keep.device)
            self._check(fn(self.h, C.c_int(map_id), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_int(r.mem),
                           Tp, C.c_void_p(ids.data_ptr()), C.c_void_p(d2.data_ptr())))
            return ids, d2
        ids = np.empty(shape, dtype=np.int32)
        d2 = np.empty(shape, dtype=r.dtype)
        self._check(fn(self.h, C.c_int(map_id), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_int(r.mem), Tp,
                       C.c_void_p(ids.ctypes.data), C.c_void_p(d2.ctypes.data)))
        return ids, d2

    def outlier_weights(self, d2):
        d2 = np.ascontiguousarray(d2)
        assert d2.dtype in (np.float32, np.float64)
        w = np.empty_like(d2)
        real = C.c_float if d2.dtype == np.float32 else C.c_double
        limit = real(0)
        nf = C.c_int(0)
        fn = getattr(self.lib, "pgicp_outlier_weights" + self._sfx(d2.dtype))
        self._check(fn(self.h, C.c_void_p(d2.ctypes.data), C.c_int(d2.shape[0]), C.c_int(HOST), C.c_void_p(w.ctypes.data),
                       C.byref(limit), C.byref(nf)))
        return w, limit.value, nf.value

    def error_stats(self, map_id, reading, ids, weights, dtype=None):
        r = _Buf(reading, dtype)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=r.dtype)
        ratio, resid = C.c_double(0), C.c_double(0)
        sys_ = (C.c_double * 30)()
        fn = getattr(self.lib, "pgicp_error_stats" + self._sfx(r.dtype))
        self._check(fn(self.h, C.c_int(map_id), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_int(r.mem),
                       C.c_void_p(ids.ctypes.data), C.c_void_p(w.ctypes.data), C.byref(ratio), C.byref(resid), sys_))
        return ratio.value, resid.value, np.array(sys_[:])

    def partial_chain(self, map_id, reading, T=None, dtype=None):
        r = _Buf(reading, dtype)
        ratio, resid = C.c_double(0), C.c_double(0)
        fn = getattr(self.lib, "pgicp_partial_chain" + self._sfx(r.dtype))
        self._check(fn(self.h, C.c_int(map_id), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_int(r.mem),
                       _T16(T) if T is not None else None, C.byref(ratio), C.byref(resid)))
        return ratio.value, resid.value

    def partial_chain_seeded(self, map_id, reading, T, src: "Context", src_start, dst_start, dtype=None):
        """pgicp_partial_chain_seeded: the partial chain with its matcher seeded from the correspondences `src`'s last align of THIS
        reading ended with; the two maps as concatenations of keyframe clouds (src_start: n_seg + 1 index boundaries in the ICP's
        map; dst_start: where each segment sits in this map, -1 = absent).  Same result as partial_chain, bit for bit."""
        r = _Buf(reading, dtype)
        ss = np.ascontiguousarray(src_start, dtype=np.int32)
        ds = np.ascontiguousarray(dst_start, dtype=np.int32)
        assert len(ss) == len(ds) + 1
        ratio, resid = C.c_double(0), C.c_double(0)
        fn = getattr(self.lib, "pgicp_partial_chain_seeded" + self._sfx(r.dtype))
        self._check(fn(self.h, C.c_int(map_id), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_int(r.n), C.c_int(r.mem),
                       _T16(T) if T is not None else None, src.h if src is not None else None, C.c_int(len(ds)),
                       C.c_void_p(ss.ctypes.data), C.c_void_p(ds.ctypes.data), C.byref(ratio), C.byref(resid)))
        return ratio.value, resid.value

    def partial_chain_batch(self, map_ids, readings, Ts, dtype=None, raise_on_error=True):
        """(weightedPointUsedRatio, residual, status) arrays of a batch of (map, reading, T) in one device pass."""
        P = len(readings)
        bufs = [_Buf(r, dtype) for r in readings]
        pa = np.zeros(P, dtype=_PROBLEM_DTYPE)              # (numpy views of the records: see align_batch)
        pa["map_id"] = map_ids
        pa["reading"] = [b.ptr for b in bufs]
        pa["stride"] = [b.stride for b in bufs]
        pa["n"] = [b.n for b in bufs]
        pa["mem"] = [b.mem for b in bufs]
        pa["T_init"] = np.asarray(Ts, dtype=np.float64).reshape(P, 16)
        ratio, resid, status = np.zeros(P, dtype=np.float64), np.zeros(P, dtype=np.float64), np.zeros(P, dtype=np.int32)
        fn = getattr(self.lib, "pgicp_partial_chain_batch" + self._sfx(bufs[0].dtype))
        rc = fn(self.h, C.c_int(P), C.c_void_p(pa.ctypes.data), C.c_void_p(ratio.ctypes.data), C.c_void_p(resid.ctypes.data),
                C.c_void_p(status.ctypes.data))
        if raise_on_error or rc not in (OK, ERR_NO_MATCH):
            self._check(rc)
        return ratio, resid, status

    def transform(self, T, pts, rotate_only=False, dtype=None):
        r = _Buf(pts, dtype)
        assert r.mem == HOST
        out = np.array(r.keep, copy=True)
        fn = getattr(self.lib, "pgicp_transform" + self._sfx(r.dtype))
        self._check(fn(self.h, _T16(T), C.c_void_p(r.ptr), C.c_int(r.stride), C.c_void_p(out.ctypes.data),
                       C.c_int(out.strides[0] // out.itemsize), C.c_int(r.n), C.c_int(int(rotate_only)), C.c_int(HOST)))
        return out

    def build_local_map(self, clouds_xyz, clouds_nrm, T_ref_kf, dtype=np.float32):
        """LocalMap::BuildCloudFromData.  numpy clouds -> numpy outputs; torch CUDA clouds (a device-resident
        keyframe cache) -> torch CUDA outputs, nothing crosses PCIe."""
        k = len(clouds_xyz)
        xs = [_Buf(c, dtype) for c in clouds_xyz]
        ns = [_Buf(c, dtype) for c in clouds_nrm]
        dtype = xs[0].dtype
        mem = xs[0].mem
        assert all(b.mem == mem and b.dtype == dtype for b in xs + ns)
        counts = (C.c_int * k)(*[b.n for b in xs])
        sx = (C.c_int * k)(*[b.stride for b in xs])
        sn = (C.c_int * k)(*[b.stride for b in ns])
        Ts = np.ascontiguousarray(np.stack([np.asarray(t, dtype=np.float64).reshape(4, 4) for t in T_ref_kf]))
        total = sum(b.n for b in xs)
        if mem == DEVICE and isinstance(clouds_xyz[0], DevPtr):
            out_x, out_n = self.device_empty(total, 3, dtype), self.device_empty(total, 3, dtype)     # (the caller frees them)
            px, pn = out_x.ptr, out_n.ptr
        elif mem == DEVICE:
            import torch
            like = clouds_xyz[0]
            out_x = torch.empty((total, 3), dtype=like.dtype, device=like.device)
            out_n = torch.empty((total, 3), dtype=like.dtype, device=like.device)
            px, pn = out_x.data_ptr(), out_n.data_ptr()
        else:
            out_x = np.zeros((total, 3), dtype=dtype)
            out_n = np.zeros((total, 3), dtype=dtype)
            px, pn = out_x.ctypes.data, out_n.ctypes.data
        PP = C.c_void_p * k
        fn = getattr(self.lib, "pgicp_build_local_map" + self._sfx(dtype))
        self._check(fn(self.h, C.c_int(k), PP(*[b.ptr for b in xs]), PP(*[b.ptr for b in ns]), sx, sn, counts,
                       C.c_void_p(Ts.ctypes.data), C.c_void_p(px), C.c_int(3), C.c_void_p(pn), C.c_int(3), C.c_int(mem)))
        return out_x, out_n

    def surface_normals(self, xyz, knn=10, max_dist=float("inf"), dtype=None, want_eigen=False, want_ids=False):
        """SurfaceNormalDataPointsFilter on the device.  numpy in -> numpy out, torch CUDA in -> torch CUDA out.
        Returns normals (n,3) [, eigenvalues (n,3) ascending] [, ids (n,knn), d2 (n,knn)]."""
        x = _Buf(xyz, dtype)
        n = x.n
        md = 1e300 if not np.isfinite(max_dist) else float(max_dist)
        if x.mem == DEVICE:
            import torch
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=xyz.device)
            nrm = mk((n, 3), xyz.dtype)
            eig = mk((n, 3), xyz.dtype) if want_eigen else None
            ids = mk((n, knn), torch.int32) if want_ids else None
            d2 = mk((n, knn), xyz.dtype) if want_ids else None
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        else:
            nrm = np.empty((n, 3), dtype=x.dtype)
            eig = np.empty((n, 3), dtype=x.dtype) if want_eigen else None
            ids = np.empty((n, knn), dtype=np.int32) if want_ids else None
            d2 = np.empty((n, knn), dtype=x.dtype) if want_ids else None
            ptr = lambda t: C.c_void_p(t.ctypes.data) if t is not None else None
        fn = getattr(self.lib, "pgicp_surface_normals" + self._sfx(x.dtype))
        self._check(fn(self.h, C.c_void_p(x.ptr), C.c_int(x.stride), C.c_int(n), C.c_int(x.mem), C.c_int(knn), C.c_double(md),
                       ptr(nrm), C.c_int(3), ptr(eig), ptr(ids), ptr(d2)))
        out = [nrm]
        if want_eigen:
            out.append(eig)
        if want_ids:
            out += [ids, d2]
        return out[0] if len(out) == 1 else tuple(out)

    def adopt_map(self, other: "Context", map_id: int) -> int:
        """Take over a map built by another context of the same device (pgicp_map_transfer)."""
        new_id = C.c_int(-1)
        self._check(self.lib.pgicp_map_transfer(other.h, C.c_int(map_id), self.h, C.byref(new_id)))
        return new_id.value

    def filter_cloud(self, filters, features, descriptors=None, T=None, rotate_rows=(-1, -1), want_idx=True):
        """pgicp_filter_cloud: `filters` = [(type, p0, p1, ...)], `features` (n, frows) host array (a point per row),
        `descriptors` (n, drows) or None.  Returns (features_out, descriptors_out, kept_idx, DevPtr of the device copy)."""
        f = np.ascontiguousarray(features)
        assert f.dtype in (np.float32, np.float64) and f.ndim == 2
        n, frows = f.shape
        d = np.ascontiguousarray(descriptors, dtype=f.dtype) if descriptors is not None else None
        drows = d.shape[1] if d is not None else 0
        fl = (Filter * max(1, len(filters)))()
        for k, spec in enumerate(filters):
            fl[k].type = int(spec[0])
            for j, v in enumerate(spec[1:]):
                fl[k].p[j] = float(v)
        of = np.empty_like(f)
        od = np.empty_like(d) if d is not None else None
        idx = np.empty(n, dtype=np.int32) if want_idx else None
        n_out = C.c_int(0)
        dev = C.c_void_p()
        fn = getattr(self.lib, "pgicp_filter_cloud" + self._sfx(f.dtype))
        self._check(fn(self.h, C.c_int(len(filters)), fl, C.c_void_p(f.ctypes.data), C.c_int(frows),
                       C.c_void_p(d.ctypes.data) if d is not None else None, C.c_int(drows), C.c_int(n), _T16(T) if T is not None else None,
                       C.c_int(rotate_rows[0]), C.c_int(rotate_rows[1]), C.c_void_p(of.ctypes.data),
                       C.c_void_p(od.ctypes.data) if od is not None else None, C.c_void_p(idx.ctypes.data) if idx is not None else None,
                       C.byref(n_out), C.byref(dev)))
        k = n_out.value
        return of[:k], (od[:k] if od is not None else None), (idx[:k] if idx is not None else None), DevPtr(dev.value, frows, k, f.dtype)

    # ---- measurement --------------------------------------------------------
    def filter_cloud_dev(self, filters, features, dropped_cap=4096):
        """pgicp_filter_cloud_dev: the device pass alone (no transformation, the host arrays untouched): (kept count, DevPtr of the
        filtered features, ascending indices of the dropped points -- None when more than dropped_cap were dropped)"""
        f = np.ascontiguousarray(features)
        assert f.ndim == 2 and f.dtype in (np.float32, np.float64)
        n, frows = f.shape
        specs = (Filter * max(1, len(filters)))()
        for k, s_ in enumerate(filters):
            specs[k].type = int(s_[0])
            for j, v in enumerate(s_[1:]):
                specs[k].p[j] = float(v)
        dropped = np.empty(max(1, dropped_cap), dtype=np.int32)
        nd, nout, dev = C.c_int(0), C.c_int(0), C.c_void_p(0)
        fn = getattr(self.lib, "pgicp_filter_cloud_dev" + self._sfx(f.dtype))
        self._check(fn(self.h, C.c_int(len(filters)), specs, C.c_void_p(f.ctypes.data), C.c_int(frows), C.c_int(n), C.c_void_p(dropped.ctypes.data),
                       C.c_int(dropped_cap), C.byref(nd), C.byref(nout), C.byref(dev)))
        return nout.value, DevPtr(dev.value, frows, nout.value, f.dtype), (dropped[:nd.value].copy() if nd.value <= dropped_cap else None)

    def profile_enable(self, on=True):
        self._check(self.lib.pgicp_profile_enable(self.h, C.c_int(int(on))))

    def debug_counters(self):
        out = (C.c_int * 4)()
        self._check(self.lib.pgicp_debug_counters(self.h, out))
        return list(out)

    def debug_last_matches(self, n, problem=0, dtype=np.float32):
        """Correspondences of the last iteration of the last align call (diagnostics; see pgicp.h)."""
        k = max(1, int(self.params.knn))
        ids = np.empty(n if k == 1 else (n, k), dtype=np.int32)
        d2 = np.empty(n if k == 1 else (n, k), dtype=dtype)
        fn = getattr(self.lib, "pgicp_debug_last_matches" + self._sfx(dtype))
        self._check(fn(self.h, C.c_int(problem), C.c_void_p(ids.ctypes.data), C.c_void_p(d2.ctypes.data)))
        return ids, d2

    def reading_order(self, n, problem=0):
        """pgicp_debug_reading_order: order[j] = index, in the caller's reading, of the point the last call sorted to position j --
        with sum_order = SUM_ORDER_SORTED the order the pairs entered the reduction tree in (hand it to the oracle as pair_order)."""
        out = np.empty(n, dtype=np.int32)
        self._check(self.lib.pgicp_debug_reading_order(self.h, C.c_int(problem), C.c_void_p(out.ctypes.data)))
        return out

    def profile_reset(self):
        self._check(self.lib.pgicp_profile_reset(self.h))

    def profile(self):
        out = {}
        for kid, name in enumerate(PROF_NAMES):
            n, ms, u, pr = C.c_longlong(0), C.c_double(0), C.c_longlong(0), C.c_longlong(0)
            self._check(self.lib.pgicp_profile_get(self.h, C.c_int(kid), C.byref(n), C.byref(ms), C.byref(u), C.byref(pr)))
            out[name] = dict(launches=n.value, total_ms=ms.value, units=u.value, problems=pr.value)
        return out


def shard_pairs(costs, world_size, rank):
    """pgicp_shard_pairs: host-only LPT split of candidate ICPs over ranks."""
    lib = load_library()
    costs = np.ascontiguousarray(costs, dtype=np.int64)
    n = costs.shape[0]
    out = np.empty(max(n, 1), dtype=np.int32)
    cnt = C.c_int(0)
    st = lib.pgicp_shard_pairs(C.c_int(n), C.c_void_p(costs.ctypes.data), C.c_int(world_size), C.c_int(rank),
                               C.c_void_p(out.ctypes.data), C.c_int(n), C.byref(cnt))
    if st != OK:
        raise PgicpError(st, "pgicp_shard_pairs")
    return out[: cnt.value].copy()


def check_icp_results(stats_records: np.ndarray, residual_error: np.ndarray, overlap_threshold=0.8, residual_error_threshold=5000.0) -> np.ndarray:
    """LoopCloser::CheckIcpResult (LoopCloser.hpp:308-340) over a record array of pgicp_stats: pgicp_check_icp_result called on
    every record in place (no per-candidate Python objects)."""
    lib = load_library()
    sa = np.ascontiguousarray(stats_records)
    assert sa.dtype == _STATS_DTYPE
    res = np.asarray(residual_error, dtype=np.float64)
    out = np.empty(len(sa), dtype=np.int32)
    base, step = sa.ctypes.data, sa.dtype.itemsize
    fn = lib.pgicp_check_icp_result
    ot, rt = C.c_double(overlap_threshold), C.c_double(residual_error_threshold)
    for k in range(len(sa)):
        out[k] = fn(C.c_void_p(base + k * step), C.c_double(res[k]), ot, rt)
    return out


def check_icp_result(stats: dict, residual_error, overlap_threshold=0.8, residual_error_threshold=5000.0) -> bool:
    """LoopCloser::CheckIcpResult (LoopCloser.hpp:308-340) through the C ABI."""
    lib = load_library()
    s = Stats()
    s.status = stats["status"]
    s.max_iter_reached = int(stats["max_iter_reached"])
    s.overlap = stats["overlap"]
    return bool(lib.pgicp_check_icp_result(C.byref(s), C.c_double(residual_error), C.c_double(overlap_threshold),
                                           C.c_double(residual_error_threshold)))
