"""Deterministic synthetic Velodyne-HDL-64E-shaped clouds.

Test / bench *input* generator only -- no ICP arithmetic lives here.  The
reference ships no sample data, so the workload of
BASELINE.json (100k-pt scan vs 1M-pt local map) is produced by this seeded
generator.  It uses a counter-based SplitMix64 stream (pure uint64 numpy
arithmetic), so every box regenerates the same bits.

World (seed 0x5EED0001): ground plane z=0, two walls y=+-8 m, 40 axis-aligned
boxes, 30 vertical cylinders.  Sensor: 64 rings, elevation +2.0..-24.8 deg,
azimuth-major firing order, height 1.73 m, max range 80 m, range noise
sigma=0.02 m.  Normals are analytic from the hit primitive, flipped toward the
sensor.  The map is assembled exactly with the semantics of
LocalMap<T>::BuildCloudFromData (reference LocalMap.hpp:209-224): every scan is
rigidly moved into the reference keyframe's frame and concatenated.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

WORLD_SEED = 0x5EED0001
NOISE_SEED = 0x5EED0002
GUESS_SEED = 0x5EED0003

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n outputs of the SplitMix64 stream `seed`, starting at counter `offset`."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 uniforms in [0,1) with 53 random bits."""
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed: int, n: int, lo: float, hi: float, offset: int = 0) -> np.ndarray:
    return lo + (hi - lo) * uniform01(seed, n, offset)


def normal01(seed: int, n: int) -> np.ndarray:
    """Box-Muller on two independent counter ranges of one stream."""
    u1 = uniform01(seed, n, 0)
    u2 = uniform01(seed, n, n)
    u1 = np.maximum(u1, 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


# ----------------------------------------------------------------------------
# SE(3) helpers (float64, column-vector convention, 4x4 homogeneous)
# ----------------------------------------------------------------------------

def rot_zyx(yaw: float, pitch: float = 0.0, roll: float = 0.0) -> np.ndarray:
    cy, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cr, sr = math.cos(roll), math.sin(roll)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return rz @ ry @ rx


def se3(x=0.0, y=0.0, z=0.0, yaw=0.0, pitch=0.0, roll=0.0) -> np.ndarray:
    t = np.eye(4)
    t[:3, :3] = rot_zyx(yaw, pitch, roll)
    t[:3, 3] = (x, y, z)
    return t


def se3_inv(t: np.ndarray) -> np.ndarray:
    r = t[:3, :3]
    out = np.eye(4)
    out[:3, :3] = r.T
    out[:3, 3] = -r.T @ t[:3, 3]
    return out


def transform_cloud(t: np.ndarray, xyz: np.ndarray, nrm: np.ndarray | None):
    """(N,3) float64 points / normals through a 4x4 rigid transform."""
    r = t[:3, :3]
    p = xyz @ r.T + t[:3, 3]
    n = None if nrm is None else nrm @ r.T
    return p, n


# ----------------------------------------------------------------------------
# world
# ----------------------------------------------------------------------------

@dataclass
class World:
    box_lo: np.ndarray   # (B,3)
    box_hi: np.ndarray   # (B,3)
    cyl_c: np.ndarray    # (C,2) centre xy
    cyl_r: np.ndarray    # (C,)
    cyl_h: np.ndarray    # (C,)
    wall_y: float = 8.0


def make_world(seed: int = WORLD_SEED, n_box: int = 40, n_cyl: int = 30) -> World:
    u = uniform01(seed, 6 * n_box + 4 * n_cyl)
    b = u[: 6 * n_box].reshape(n_box, 6)
    size = 1.0 + 5.0 * b[:, 0:3]
    size[:, 2] = 1.0 + 3.0 * b[:, 2]              # heights 1..4 m
    cx = -60.0 + 120.0 * b[:, 3]
    cy = -7.0 + 14.0 * b[:, 4]
    # keep a drivable corridor |y| < 2 free of boxes
    cy = np.where(np.abs(cy) < 3.5, np.sign(cy + 1e-9) * (3.5 + np.abs(cy)), cy)
    size[:, 1] = np.minimum(size[:, 1], 3.0)
    lo = np.stack([cx - size[:, 0] / 2, cy - size[:, 1] / 2, np.zeros(n_box)], 1)
    hi = np.stack([cx + size[:, 0] / 2, cy + size[:, 1] / 2, size[:, 2]], 1)
    c = u[6 * n_box:].reshape(n_cyl, 4)
    ccx = -60.0 + 120.0 * c[:, 0]
    ccy = -7.0 + 14.0 * c[:, 1]
    ccy = np.where(np.abs(ccy) < 2.5, np.sign(ccy + 1e-9) * (2.5 + np.abs(ccy)), ccy)
    return World(lo, hi, np.stack([ccx, ccy], 1), 0.15 + 0.25 * c[:, 2], 2.0 + 4.0 * c[:, 3])


def _raycast(world: World, o: np.ndarray, d: np.ndarray, max_range: float):
    """Nearest hit of rays o + t d (d unit, (R,3)) -> t (R,), normal (R,3)."""
    n_rays = d.shape[0]
    best_t = np.full(n_rays, np.inf)
    best_n = np.zeros((n_rays, 3))

    def take(t, nrm):
        nonlocal best_t, best_n
        m = (t > 1e-6) & (t < best_t)
        best_t = np.where(m, t, best_t)
        best_n = np.where(m[:, None], nrm, best_n)

    with np.errstate(divide="ignore", invalid="ignore"):
        # ground z = 0
        t = np.where(d[:, 2] < 0, -o[2] / d[:, 2], np.inf)
        take(t, np.broadcast_to(np.array([0, 0, 1.0]), (n_rays, 3)))
        # walls y = +-wall_y
        for s in (+1.0, -1.0):
            t = np.where(s * d[:, 1] > 0, (s * world.wall_y - o[1]) / d[:, 1], np.inf)
            take(t, np.broadcast_to(np.array([0, -s, 0.0]), (n_rays, 3)))
        # boxes (slab method), vectorised over rays per box
        inv = 1.0 / d
        for lo, hi in zip(world.box_lo, world.box_hi):
            t0 = (lo - o) * inv
            t1 = (hi - o) * inv
            tn = np.minimum(t0, t1)
            tf = np.maximum(t0, t1)
            tn = np.where(np.isnan(tn), -np.inf, tn)
            tf = np.where(np.isnan(tf), np.inf, tf)
            axis = np.argmax(tn, axis=1)
            tnear = np.max(tn, axis=1)
            tfar = np.min(tf, axis=1)
            hit = (tnear <= tfar) & (tnear > 1e-6)
            nrm = np.zeros((n_rays, 3))
            sgn = -np.sign(d[np.arange(n_rays), axis])
            nrm[np.arange(n_rays), axis] = sgn
            take(np.where(hit, tnear, np.inf), nrm)
        # vertical cylinders
        a = d[:, 0] ** 2 + d[:, 1] ** 2
        for (cx, cy), r, h in zip(world.cyl_c, world.cyl_r, world.cyl_h):
            ox, oy = o[0] - cx, o[1] - cy
            bq = ox * d[:, 0] + oy * d[:, 1]
            cq = ox * ox + oy * oy - r * r
            disc = bq * bq - a * cq
            t = (-bq - np.sqrt(np.maximum(disc, 0))) / a
            z = o[2] + t * d[:, 2]
            hit = (disc > 0) & (a > 1e-12) & (z >= 0) & (z <= h)
            px = ox + t * d[:, 0]
            py = oy + t * d[:, 1]
            nrm = np.stack([px / r, py / r, np.zeros(n_rays)], 1)
            take(np.where(hit, t, np.inf), nrm)
    ok = best_t <= max_range
    return best_t, best_n, ok


def make_scan(world: World, T_world_sensor: np.ndarray, n_points: int, scan_idx: int,
              rings: int = 64, az_steps: int | None = None, max_range: float = 80.0,
              noise_sigma: float = 0.02, height: float = 1.73):
    """One scan in the SENSOR frame: (xyz float32 (N,3), normals float32 (N,3)).

    `T_world_sensor` is the robot pose on the ground; the sensor sits `height`
    above it.  Exactly `n_points` returns are kept (deterministic uniform
    subsample of the valid returns, in firing order).
    """
    if az_steps is None:
        az_steps = int(math.ceil(n_points / rings * 1.6))
    t_ws = T_world_sensor @ se3(z=height)
    while True:
        elev = np.deg2rad(np.linspace(2.0, -24.8, rings))
        az = np.arange(az_steps) * (2.0 * math.pi / az_steps)
        azg, elg = np.meshgrid(az, elev, indexing="ij")          # azimuth-major
        azg, elg = azg.ravel(), elg.ravel()
        d_s = np.stack([np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], 1)
        d_w = d_s @ t_ws[:3, :3].T
        t, n_w, ok = _raycast(world, t_ws[:3, 3], d_w, max_range)
        if int(ok.sum()) >= n_points:
            break
        az_steps = int(az_steps * 1.3) + 1
    idx = np.nonzero(ok)[0]
    sel = idx[(np.arange(n_points, dtype=np.int64) * idx.size) // n_points]
    noise = noise_sigma * normal01(NOISE_SEED + scan_idx, d_s.shape[0])
    rng = t[sel] + noise[sel]
    xyz_s = d_s[sel] * rng[:, None]
    n_s = n_w[sel] @ t_ws[:3, :3]                                  # world -> sensor (R^T n)
    flip = np.sum(n_s * d_s[sel], axis=1) > 0                      # face the sensor
    n_s = np.where(flip[:, None], -n_s, n_s)
    # express in the ROBOT frame (sensor mounted `height` above the robot origin);
    # this is what Localizer::ProcessData does at Localizer.hpp:106.
    xyz_r = xyz_s + np.array([0.0, 0.0, height])
    return xyz_r.astype(np.float32), n_s.astype(np.float32)


# ----------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------

@dataclass
class ScanToMap:
    map_xyz: np.ndarray        # (M,3) f32, reference-keyframe frame
    map_nrm: np.ndarray        # (M,3) f32
    scans_xyz: list            # B x (N,3) f32, robot frame of each query pose
    scans_nrm: list
    T_truth: list              # B x 4x4 f64: T_refkf_robot (ground truth)
    T_init: list               # B x 4x4 f64: perturbed initial guess


def perturbation(scan_idx: int) -> np.ndarray:
    u = uniform01(GUESS_SEED + scan_idx, 6)
    dt = -0.3 + 0.6 * u[0:3]
    yaw = math.radians(-2.0 + 4.0 * u[3])
    roll = math.radians(-0.5 + 1.0 * u[4])
    pitch = math.radians(-0.5 + 1.0 * u[5])
    return se3(dt[0], dt[1], dt[2], yaw, pitch, roll)


def make_scan_to_map(n_scan: int = 100_000, n_map: int = 1_000_000, n_queries: int = 64,
                     n_map_poses: int = 12, rings: int = 64, spacing: float = 1.5) -> ScanToMap:
    """BASELINE.json configs[1] (and, with 24 poses / 2M, configs[2])."""
    world = make_world()
    jitter = np.deg2rad(uniform(WORLD_SEED + 17, n_map_poses, -2.0, 2.0))
    poses = [se3(x=spacing * i, yaw=float(jitter[i])) for i in range(n_map_poses)]
    T_ref = poses[-1]
    T_ref_inv = se3_inv(T_ref)
    per_scan = max(n_scan, -(-n_map // n_map_poses))
    parts_p, parts_n = [], []
    # reference keyframe first, then newest -> oldest (LocalMap.hpp:213-223)
    for i in [n_map_poses - 1] + list(range(n_map_poses - 2, -1, -1)):
        xyz, nrm = make_scan(world, poses[i], per_scan, 1000 + i, rings=rings)
        p, n = transform_cloud(T_ref_inv @ poses[i], xyz.astype(np.float64), nrm.astype(np.float64))
        parts_p.append(p)
        parts_n.append(n)
    mp = np.concatenate(parts_p)
    mn = np.concatenate(parts_n)
    sel = (np.arange(n_map, dtype=np.int64) * mp.shape[0]) // n_map
    mp, mn = mp[sel].astype(np.float32), mn[sel].astype(np.float32)

    qu = uniform01(WORLD_SEED + 33, 3 * n_queries).reshape(n_queries, 3)
    scans_p, scans_n, truth, init = [], [], [], []
    for b in range(n_queries):
        dx = 0.75 + (-0.5 + 1.0 * qu[b, 0])
        dy = -0.3 + 0.6 * qu[b, 1]
        dyaw = math.radians(-3.0 + 6.0 * qu[b, 2])
        T_q = T_ref @ se3(x=dx, y=dy, yaw=dyaw)
        xyz, nrm = make_scan(world, T_q, n_scan, b, rings=rings)
        T_true = T_ref_inv @ T_q
        scans_p.append(xyz)
        scans_n.append(nrm)
        truth.append(T_true)
        init.append(T_true @ perturbation(b))
    return ScanToMap(mp, mn, scans_p, scans_n, truth, init)


@dataclass
class PairSet:
    reading_xyz: list   # P x (N,3)
    ref_xyz: list       # P x (M,3)
    ref_nrm: list       # P x (M,3)
    T_truth: list
    T_init: list


def make_pairs(n_pairs: int, n_pts: int = 100_000, n_keyframes: int = 24, rings: int = 64,
               spacing: float = 1.0, first_pair: int = 0) -> PairSet:
    """BASELINE.json configs[4]: loop-closure candidate pairs (keyframe i vs j,
    |dt| <= 3 m -- the geometric threshold of LoopCloser.hpp:17)."""
    world = make_world()
    jitter = np.deg2rad(uniform(WORLD_SEED + 51, n_keyframes, -3.0, 3.0))
    poses = [se3(x=-12.0 + spacing * i, yaw=float(jitter[i])) for i in range(n_keyframes)]
    cache = {}

    def kf(i):
        if i not in cache:
            cache[i] = make_scan(world, poses[i], n_pts, 2000 + i, rings=rings)
        return cache[i]

    out = PairSet([], [], [], [], [])
    for p in range(first_pair, first_pair + n_pairs):
        i = p % n_keyframes
        j = (i + 1 + (p // n_keyframes) % 3) % n_keyframes
        if abs(i - j) * spacing > 3.0:
            j = (i + 1) % n_keyframes
            if abs(i - j) * spacing > 3.0:
                j = i - 1
        rd_xyz, _ = kf(j)
        rf_xyz, rf_nrm = kf(i)
        T_true = se3_inv(poses[i]) @ poses[j]
        out.reading_xyz.append(rd_xyz)
        out.ref_xyz.append(rf_xyz)
        out.ref_nrm.append(rf_nrm)
        out.T_truth.append(T_true)
        out.T_init.append(T_true @ perturbation(5000 + p))
    return out


def make_two_scans(n_pts: int = 10_000, rings: int = 16):
    """BASELINE.json configs[0]: two 10k-pt scans 0.5 m apart."""
    world = make_world()
    T_a = se3(x=0.0)
    T_b = se3(x=0.5, yaw=math.radians(1.0))
    ref_xyz, ref_nrm = make_scan(world, T_a, n_pts, 3000, rings=rings)
    rd_xyz, rd_nrm = make_scan(world, T_b, n_pts, 3001, rings=rings)
    T_true = se3_inv(T_a) @ T_b
    return dict(ref_xyz=ref_xyz, ref_nrm=ref_nrm, reading_xyz=rd_xyz, reading_nrm=rd_nrm,
                T_truth=T_true, T_init=T_true @ perturbation(9000))


@dataclass
class Drive:
    poses_true: list    # S x 4x4 T_world_robot
    odom: list          # S x 4x4 odometry poses (true increments + drift)
    scans_xyz: list     # S x (N,3) robot frame
    scans_nrm: list


DRIVE_SEED = 0x5EED0004


def make_drive(n_scans: int, n_pts: int = 100_000, step: float = 0.35, rings: int = 64, x0: float = -40.0,
               odom_sigma_t: float = 0.02, odom_sigma_yaw_deg: float = 0.15) -> Drive:
    """BASELINE.json configs[2]: a 10 Hz feed along the street (`step` m per scan, gentle weaving);
    the odometry reports every true increment with a small error, so its pose drifts and the ICP
    has something to correct (Localizer.hpp:119-127)."""
    world = make_world()
    u = uniform01(DRIVE_SEED, 3 * n_scans)
    poses, odom = [], []
    T_o = None
    for s in range(n_scans):
        yaw = math.radians(3.0) * math.sin(0.15 * s)
        T = se3(x=x0 + step * s, y=0.8 * math.sin(0.05 * s), yaw=yaw)
        if s == 0:
            T_o = T.copy()
        else:
            d_true = se3_inv(poses[-1]) @ T
            err = se3(x=odom_sigma_t * (2 * u[3 * s] - 1), y=odom_sigma_t * (2 * u[3 * s + 1] - 1),
                      yaw=math.radians(odom_sigma_yaw_deg) * (2 * u[3 * s + 2] - 1))
            T_o = T_o @ d_true @ err
        poses.append(T)
        odom.append(T_o.copy())
    scans = [make_scan(world, poses[s], n_pts, 7000 + s, rings=rings) for s in range(n_scans)]
    return Drive(poses, odom, [c[0] for c in scans], [c[1] for c in scans])


# ----------------------------------------------------------------------------
# BASELINE.json configs[3]: a KITTI-00-shaped sequence -- a long drive through a street grid that comes back
# along streets it has already driven, so that a pose-graph SLAM front end finds loop closures
# ----------------------------------------------------------------------------

CITY_SEED = 0x5EED0005

# The route on the street grid, block by block: an outer loop, the first street again (first loop closures), then a
# second loop through the middle that re-drives parts of the first in both directions (14 block edges, 4 of them twice).
CITY_ROUTE = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 0), (1, 0), (1, 1), (1, 2), (2, 2), (2, 1), (1, 1), (1, 0), (2, 0)]


@dataclass
class City:
    block: float         # street spacing, metres
    box_lo: np.ndarray   # buildings (B,3)
    box_hi: np.ndarray
    cyl_c: np.ndarray    # trees / poles (C,2)
    cyl_r: np.ndarray
    cyl_h: np.ndarray

    def near(self, x: float, y: float, reach: float = 95.0) -> World:
        """The part of the city a sensor at (x, y) can see, as a World (no street-canyon walls)."""
        bc = 0.5 * (self.box_lo[:, :2] + self.box_hi[:, :2])
        half = 0.5 * np.linalg.norm(self.box_hi[:, :2] - self.box_lo[:, :2], axis=1)
        mb = np.hypot(bc[:, 0] - x, bc[:, 1] - y) - half < reach
        mc = np.hypot(self.cyl_c[:, 0] - x, self.cyl_c[:, 1] - y) < reach
        return World(self.box_lo[mb], self.box_hi[mb], self.cyl_c[mc], self.cyl_r[mc], self.cyl_h[mc], wall_y=1e9)


def make_city(block: float, seed: int = CITY_SEED) -> City:
    """Buildings line both sides of every street of a 2 x 2-block grid (3 x 3 streets); trees and poles stand
    between them and the roadway.  Deterministic; sizes drawn from the SplitMix64 stream."""
    streets = []                                   # (x0, y0, x1, y1) centre lines
    for k in range(3):
        streets.append((0.0, k * block, 2 * block, k * block))
        streets.append((k * block, 0.0, k * block, 2 * block))
    lo, hi, cc, cr, chh = [], [], [], [], []
    u = uniform01(seed, 400000)
    p = 0
    for (x0, y0, x1, y1) in streets:
        horizontal = y0 == y1
        length = (x1 - x0) if horizontal else (y1 - y0)
        for side in (-1.0, 1.0):
            s = -40.0                              # buildings continue a little beyond the grid's ends
            while s < length + 40.0:
                blen = 8.0 + 17.0 * u[p]; gap = 2.0 + 6.0 * u[p + 1]; depth = 8.0 + 6.0 * u[p + 2]
                height = 4.0 + 8.0 * u[p + 3]; setback = 7.0 + 2.5 * u[p + 4]
                p += 5
                a, b = s, s + blen
                s = b + gap
                # leave the crossings open
                mid = 0.5 * (a + b)
                if any(abs(mid - k * block) < 0.5 * blen + 10.0 for k in range(3)):
                    continue
                n0, n1 = side * setback, side * (setback + depth)
                if horizontal:
                    lo.append([x0 + a, y0 + min(n0, n1), 0.0]); hi.append([x0 + b, y0 + max(n0, n1), height])
                else:
                    lo.append([x0 + min(n0, n1), y0 + a, 0.0]); hi.append([x0 + max(n0, n1), y0 + b, height])
            s = 5.0
            while s < length - 5.0:                # trees and poles, 4.5-6 m from the centre line
                spacing = 9.0 + 14.0 * u[p]; off = side * (4.5 + 1.5 * u[p + 1]); r = 0.12 + 0.25 * u[p + 2]; h = 3.0 + 5.0 * u[p + 3]
                p += 4
                if not any(abs(s - k * block) < 9.0 for k in range(3)):
                    cc.append([x0 + s, y0 + off] if horizontal else [x0 + off, y0 + s]); cr.append(r); chh.append(h)
                s += spacing
    return City(block, np.array(lo), np.array(hi), np.array(cc), np.array(cr), np.array(chh))


def city_route(n_scans: int, step: float, lane_jitter: float = 0.6, corner_radius: float = 7.0):
    """`n_scans` poses along CITY_ROUTE, `step` metres apart (arc length), with rounded corners; a street that is
    driven twice is driven `lane_jitter` metres to the side the second time.  Returns (block, [T_world_robot])."""
    edges = len(CITY_ROUTE) - 1
    block = max(40.0, ((n_scans - 1) * step + 6.0 * edges + 10.0) / edges)      # (corner rounding shortens the path: ~4 m per turn)
    pts = []
    seen = {}
    for k, (ix, iy) in enumerate(CITY_ROUTE):
        pts.append(np.array([ix * block, iy * block], dtype=np.float64))
    # lateral offset per edge: second traversal of the same street segment shifts sideways
    offs = []
    for k in range(edges):
        key = tuple(sorted([CITY_ROUTE[k], CITY_ROUTE[k + 1]]))
        offs.append(lane_jitter * seen.get(key, 0))
        seen[key] = seen.get(key, 0) + 1
    # polyline with offsets applied perpendicular to each edge; corners become arcs by sampling a smoothed path
    fine = []
    for k in range(edges):
        a, b = pts[k], pts[k + 1]
        d = (b - a) / np.linalg.norm(b - a)
        nrm = np.array([-d[1], d[0]])
        a2, b2 = a + offs[k] * nrm, b + offs[k] * nrm
        m = max(2, int(np.linalg.norm(b2 - a2) / 0.25))
        t = np.linspace(0.0, 1.0, m, endpoint=False)
        fine.append(a2[None, :] + t[:, None] * (b2 - a2)[None, :])
    fine.append((pts[-1] + offs[-1] * np.array([0.0, 0.0]))[None, :])
    path = np.concatenate(fine)
    # corner rounding: moving average over 2*corner_radius of arc length
    w = max(1, int(2 * corner_radius / 0.25))
    pad = np.concatenate([np.repeat(path[:1], w, 0), path, np.repeat(path[-1:], w, 0)])
    ker = np.ones(2 * w + 1) / (2 * w + 1)
    sm = np.stack([np.convolve(pad[:, 0], ker, mode="same"), np.convolve(pad[:, 1], ker, mode="same")], 1)[w:-w]
    seg = np.linalg.norm(np.diff(sm, axis=0), axis=1)
    arc = np.concatenate([[0.0], np.cumsum(seg)])
    want = np.minimum(np.arange(n_scans) * step, arc[-1] - 1e-6)
    xs = np.interp(want, arc, sm[:, 0]); ys = np.interp(want, arc, sm[:, 1])
    ahead = np.minimum(want + 1.0, arc[-1])
    hx = np.interp(ahead, arc, sm[:, 0]) - xs; hy = np.interp(ahead, arc, sm[:, 1]) - ys
    yaw = np.arctan2(hy, hx)
    yaw[-1] = yaw[-2] if n_scans > 1 else 0.0
    return block, [se3(x=float(xs[s]), y=float(ys[s]), yaw=float(yaw[s])) for s in range(n_scans)]


def city_odometry(poses, sigma_t: float = 0.02, sigma_yaw_deg: float = 0.1, seed: int = CITY_SEED + 1):
    """Odometry poses: every true increment with a small uniform error, accumulated (drifts)."""
    u = uniform01(seed, 3 * len(poses))
    odom = [poses[0].copy()]
    for s in range(1, len(poses)):
        d_true = se3_inv(poses[s - 1]) @ poses[s]
        err = se3(x=sigma_t * (2 * u[3 * s] - 1), y=sigma_t * (2 * u[3 * s + 1] - 1),
                  yaw=math.radians(sigma_yaw_deg) * (2 * u[3 * s + 2] - 1))
        odom.append(odom[-1] @ d_true @ err)
    return odom


def make_city_scan(city: City, T_world_robot: np.ndarray, n_pts: int, scan_idx: int, rings: int = 16):
    return make_scan(city.near(float(T_world_robot[0, 3]), float(T_world_robot[1, 3])), T_world_robot, n_pts, 9000 + scan_idx, rings=rings)
