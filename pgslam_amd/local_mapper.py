"""Streaming local mapper: pgslam's Localizer + LocalMap on a device-resident sliding map.

Reference behaviour (paths relative to /root/reference/src/pgslam/):
  * Localizer::ProcessData (Localizer.hpp:91-135): first cloud -> first keyframe and map
    (ProcessFirstCloud :137-152); afterwards the odometry delta gives the initial guess
    `T_refkf_robot * (last_odom^-1 * odom)` (:119-123), ICP corrects it (:126), the world pose
    is `T_world_refkf * T_refkf_robot` (:127) and UpdateAfterIcp (:178-268) decides what the next
    local map is.
  * UpdateAfterIcp, the graph-free part: overlap >= threshold keeps the keyframe set and makes the
    keyframe closest to the robot the reference (case #2, :213-221); otherwise the scan becomes a
    new keyframe appended to the circular composition (:236-246).  A changed composition rebuilds
    the cloud and calls setMap (:254-266), and a changed reference re-expresses the robot pose
    (UpdateRefkfRobotPose :271-274).
  * LocalMap::BuildCloudFromData (LocalMap.hpp:209-224): the reference keyframe (back of the
    circular buffer) is copied, every other keyframe -- newest to oldest -- is moved by
    `T_refkf_world * T_world_kf` and concatenated.
The neighbour-composition search (FindNeighborLocalMapComposition, Localizer.hpp:393-483) needs the
pose graph and is out of this module's scope.

MI355X design: keyframe clouds stay in HBM (torch CUDA tensors); a new map is assembled by
pgicp_build_local_map and indexed by pgicp_map_create entirely on the device.  With
`async_rebuild` a second context (own HIP stream and scratch) does that in a background thread
while scans keep aligning against the current map; the finished index is handed over with
pgicp_map_transfer and the pose is re-expressed in the new reference frame.  The synchronous mode
reproduces the reference's order of operations exactly and is what the parity tests use.

Host clouds (the caller owns the scans, Localizer.hpp:103-126; LocalizerMT.hpp:27-40 has the next scan queued while the
current one aligns): `process(odom, xyz, nrm, next_xyz=...)` with numpy scans starts the upload of the NEXT scan on the
context's copy stream (pgicp_upload_*) before it aligns the current one, so the transfer hides behind the ICP; the
alignment is handed the device pointer of an upload that was started one call earlier.  A scan that becomes a keyframe is
copied into device memory of its own (the upload sets are recycled two uploads later).
"""
from __future__ import annotations

import threading
from collections import deque
from dataclasses import dataclass, field

import numpy as np

from . import icp


@dataclass
class LocalMapperConfig:
    capacity: int = 3                    # keyframes in the local map (LocalMap capacity)
    overlap_threshold: float = 0.8       # Localizer.hpp:27
    minimal_overlap: float = 0.5         # Localizer.hpp:28 (warning level only)
    chain: dict = field(default_factory=dict)
    async_rebuild: bool = False


@dataclass
class Keyframe:
    id: int
    xyz: object                          # (N,3) cloud in the robot frame at capture (Localizer.hpp:106,239)
    nrm: object
    T_world_kf: np.ndarray


def _inv(T):
    R, t = T[:3, :3], T[:3, 3]
    out = np.eye(4)
    out[:3, :3] = R.T
    out[:3, 3] = -R.T @ t
    return out


def composition_transforms(window):
    """Order and transforms of LocalMap::BuildCloudFromData: reference (back) first with identity,
    then newest -> oldest with T_refkf_world * T_world_kf."""
    ref = window[-1]
    T_ref_world = _inv(ref.T_world_kf)
    order = [ref] + list(reversed(list(window)[:-1]))
    return order, [np.eye(4)] + [T_ref_world @ kf.T_world_kf for kf in order[1:]]


class StreamingLocalMapper:
    """`backend` is an icp.Context (the product path) or any object with the same set_map / align /
    build_local_map / destroy_map methods (the tests plug the CPU oracle in here)."""

    def __init__(self, backend, cfg: LocalMapperConfig, builder=None, to_device=None):
        self.be = backend
        self.cfg = cfg
        self.builder = builder if builder is not None else backend
        if cfg.async_rebuild and builder is None:
            raise ValueError("async_rebuild needs a second context (builder) with its own stream")
        self.to_device = to_device or (lambda a: a)
        if cfg.chain and hasattr(backend, "set_params"):
            backend.set_params(**cfg.chain)
            if builder is not None:
                builder.set_params(**cfg.chain)
        self.window = deque(maxlen=cfg.capacity)
        self.map_id = None
        self.map_window = None               # keyframes of the map being served, reference last
        self.T_refkf_robot = np.eye(4)
        self.T_world_robot = np.eye(4)
        self.last_odom = None
        self.next_kf_id = 0
        self.count = 0
        self.keyframe_scans = []
        self.rebuilds = 0
        self.last_stats = None
        self._pending = None                 # (thread, result holder, window snapshot)
        self._job = None                     # scan between prepare() and complete()
        self._staged = None                  # (host array, device handle) of a scan uploaded one call ahead
        self.pinned_sources = False          # host scans lie in pinned memory (pgicp_host_alloc): no staging copy

    # ---- map (re)building ---------------------------------------------------------------
    def _build(self, ctx, window):
        order, Ts = composition_transforms(window)
        xyz, nrm = ctx.build_local_map([k.xyz for k in order], [k.nrm for k in order], Ts)
        return ctx.set_map(xyz, nrm, center=True)

    def _install(self, map_id, window):
        old = self.map_id
        self.map_id, self.map_window = map_id, list(window)
        if old is not None:
            self.be.destroy_map(old)
        self.rebuilds += 1

    def _rebuild(self, old_ref):
        window = list(self.window)
        if not self.cfg.async_rebuild:
            self._install(self._build(self.be, window), window)
            if self.window[-1] is not old_ref:
                self.T_refkf_robot = _inv(self.window[-1].T_world_kf) @ self.T_world_robot   # Localizer.hpp:273
            return
        self._finish_pending(wait=True)
        holder = {}

        def work():
            try:
                holder["id"] = self._build(self.builder, window)
            except Exception as e:                  # surfaced on the caller's thread
                holder["err"] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()                                  # ctypes releases the GIL inside the library
        self._pending = (th, holder, window)

    def _finish_pending(self, wait=False):
        if self._pending is None:
            return
        th, holder, window = self._pending
        if th.is_alive() and not wait:
            return
        th.join()
        self._pending = None
        if "err" in holder:
            raise holder["err"]
        self._install(self.be.adopt_map(self.builder, holder["id"]), window)
        # the pose tracked against the previous map, re-expressed in the new reference keyframe
        self.T_refkf_robot = _inv(window[-1].T_world_kf) @ self.T_world_robot

    # ---- per scan -----------------------------------------------------------------------
    def process(self, odom_T_world_robot, scan_xyz, scan_nrm, next_xyz=None):
        """One scan (already in the robot frame).  Returns T_world_robot.  `next_xyz`: the host scan that will be
        processed next -- its upload starts now and travels while this scan aligns."""
        job = self.prepare(odom_T_world_robot, scan_xyz, scan_nrm)
        if next_xyz is not None:
            self.stage(next_xyz)
        if job is not None:
            T, stats = self.be.align(job[0], job[1], job[2])                  # Localizer.hpp:126
            self.complete(T, stats)
        return self.T_world_robot.copy()

    # ---- host scans, one step ahead ------------------------------------------------------
    def _is_host(self, a):
        return isinstance(a, np.ndarray) and hasattr(self.be, "upload")

    def stage(self, host_xyz):
        """Start the transfer of a host scan on the context's copy stream (returns at once)."""
        if self._is_host(host_xyz):
            self._staged = (host_xyz, self.be.upload([host_xyz], pinned=self.pinned_sources)[0])

    def _reading(self, scan_xyz):
        """What the ICP call is handed: the device pointer of the upload started one call ago (or started now)."""
        if not self._is_host(scan_xyz):
            return self.to_device(scan_xyz)
        if self._staged is None or self._staged[0] is not scan_xyz:
            self.stage(scan_xyz)
        handle = self._staged[1]
        self._staged = None
        return handle

    def _own(self, host_array):
        """A keyframe's cloud must outlive the upload sets (recycled two uploads later): device memory of its own."""
        import torch
        return torch.from_numpy(np.ascontiguousarray(host_array)).to(torch.device("cuda", self.be.device))

    def prepare(self, odom_T_world_robot, scan_xyz, scan_nrm):
        """First half of ProcessData: everything up to the ICP call.  Returns (map_id, reading, T_init) for
        the caller to align -- alone or as one problem of a device batch (StreamingFleet) -- or None when
        the scan was consumed as the first keyframe."""
        odom = np.asarray(odom_T_world_robot, dtype=np.float64).reshape(4, 4)
        self.count += 1
        if self.map_id is None and self._pending is None:
            if self._is_host(scan_xyz):
                scan_xyz, scan_nrm = self._own(scan_xyz), self._own(scan_nrm)
            kf = Keyframe(self.next_kf_id, self.to_device(scan_xyz), self.to_device(scan_nrm), odom.copy())
            self.next_kf_id += 1
            self.window.append(kf)
            self.keyframe_scans.append(self.count - 1)
            self._install(self._build(self.be, list(self.window)), list(self.window))
            self.T_refkf_robot = np.eye(4)
            self.T_world_robot = odom.copy()
            self.last_odom = odom.copy()
            return None
        self._finish_pending(wait=False)
        d_odom = _inv(self.last_odom) @ odom                                   # Localizer.hpp:119
        T_init = self.T_refkf_robot @ d_odom                                   # :123
        self._job = (odom, self._reading(scan_xyz), scan_nrm, scan_xyz)
        return self.map_id, self._job[1], T_init

    def complete(self, T, stats):
        """Second half of ProcessData: the ICP result, the world pose, UpdateAfterIcp."""
        odom, dev_xyz, scan_nrm, host_xyz = self._job
        self._job = None
        self.last_stats = stats
        self.T_refkf_robot = np.asarray(T, dtype=np.float64).reshape(4, 4)
        self.T_world_robot = self.map_window[-1].T_world_kf @ self.T_refkf_robot   # :127
        self._update_after_icp(stats["overlap"], dev_xyz, scan_nrm, host_xyz)
        self.last_odom = odom.copy()

    def _update_after_icp(self, overlap, dev_xyz, scan_nrm, host_xyz=None):
        if self._pending is not None:
            return                      # a rebuild is in flight: decisions resume on the new map
        old_ref = self.window[-1]
        changed = False
        if overlap >= self.cfg.overlap_threshold:
            # case #2 (Localizer.hpp:213-221): reference := keyframe closest to the robot
            # (one vectorised pass over the window's positions: twenty np.linalg.norm calls were 80 us of a 930 us scan)
            d = np.array([k.T_world_kf[:3, 3] for k in self.window]) - self.T_world_robot[:3, 3]
            closest = int(np.argmin(np.einsum("ij,ij->i", d, d)))   # first minimum, as FindClosestVertex (LocalMap.hpp:185-203)
            if self.window[closest] is not old_ref:
                items = list(self.window)
                items[closest], items[-1] = items[-1], items[closest]   # std::iter_swap (:220)
                self.window = deque(items, maxlen=self.cfg.capacity)
                changed = True
        else:
            if isinstance(dev_xyz, icp.DevPtr):          # a host scan: the keyframe owns device copies of cloud and normals
                dev_xyz, scan_nrm = self._own(host_xyz), self._own(scan_nrm)
            kf = Keyframe(self.next_kf_id, dev_xyz, self.to_device(scan_nrm), self.T_world_robot.copy())
            self.next_kf_id += 1
            self.window.append(kf)                      # circular buffer: the oldest drops out
            self.keyframe_scans.append(self.count - 1)
            changed = True
        if changed:
            self._rebuild(old_ref)

    def close(self):
        self._finish_pending(wait=True)
        if self.map_id is not None:
            self.be.destroy_map(self.map_id)
            self.map_id = None


class StreamingFleet:
    """Several independent vehicles on one GPU, stepped together: the scans of one time step are aligned
    as ONE device batch against each vehicle's own map (pgicp_align_batch), which fills the GPU where a
    single 100k-point scan cannot (a chain of small kernels).  Every vehicle keeps its own mapper state
    and takes exactly the decisions it would take alone; only the ICP calls are shared."""

    def __init__(self, backend, n_vehicles, cfg: LocalMapperConfig, builder=None, to_device=None):
        if cfg.async_rebuild:
            raise ValueError("StreamingFleet rebuilds maps in line (one context serves all vehicles)")
        self.be = backend
        self.mappers = [StreamingLocalMapper(backend, cfg, builder=builder, to_device=to_device) for _ in range(n_vehicles)]

    def step(self, odoms, scans_xyz, scans_nrm):
        # host scans of one time step travel as ONE upload (a context keeps two upload sets: one in use, one in flight;
        # a third separate upload would overwrite the first before the batch has read it)
        host = [k for k, (m, x) in enumerate(zip(self.mappers, scans_xyz)) if m._is_host(x) and m.map_id is not None]
        if host:
            handles = self.be.upload([scans_xyz[k] for k in host], pinned=self.mappers[0].pinned_sources)
            for k, h in zip(host, handles):
                self.mappers[k]._staged = (scans_xyz[k], h)
        jobs = [m.prepare(o, x, n) for m, o, x, n in zip(self.mappers, odoms, scans_xyz, scans_nrm)]
        live = [k for k, j in enumerate(jobs) if j is not None]
        if live:
            Ts, stats = self.be.align_batch([jobs[k][0] for k in live], [jobs[k][1] for k in live], [jobs[k][2] for k in live])
            for n, k in enumerate(live):
                self.mappers[k].complete(Ts[n], stats[n])
        return [m.T_world_robot.copy() for m in self.mappers]

    def close(self):
        for m in self.mappers:
            m.close()
