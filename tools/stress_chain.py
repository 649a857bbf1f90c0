"""Randomised stress of the chain's OTHER modules (GPU box; round 4): as tools/stress_batch.py -- batches of 1-16 problems over
1-3 maps of random scenes, whole ICP runs against the CPU oracle -- with a random chain per batch: KDTreeMatcher.knn 1-4,
PointToPlane / PointToPoint / PointToPlane{force4DOF}, a SurfaceNormalOutlierFilter (reading normals from the scan generator, a
share of them turned), a BoundTransformationChecker (loose or tight: PGICP_ERR_BOUND must come from both sides), MedianDist or
TrimmedDist, a MaxDistOutlierFilter; three batches of ten with double scalars against the double oracle.  Status, iteration count, n_finite, n_kept, threshold (bit for bit in float), transform to
1e-5 m / 1e-5 rad.  tools/stress_chain.py [seconds] [seed]"""
import sys, time, importlib
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from pgslam_amd import icp, synth
orc = importlib.import_module("oracle.oracle")
CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o32 = orc.Oracle(np.float32)
o64 = orc.Oracle(np.float64)
world = synth.make_world()


def pose_error(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    return float(np.linalg.norm(d[:3, 3])), float(np.linalg.norm([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) / 2.0)


def scene():
    x0 = float(rng.uniform(-45, 20))
    poses = [synth.se3(x=x0 + 2.0 * k) for k in range(int(rng.integers(1, 4)))]
    ref_inv = synth.se3_inv(poses[0])
    parts = []
    for P in poses:
        x, nn = synth.make_scan(world, P, int(rng.integers(2000, 6000)), int(rng.integers(1, 1 << 30)), rings=16, max_range=14.0)
        parts.append(synth.transform_cloud(ref_inv @ P, x.astype(np.float64), nn.astype(np.float64)))
    ref = np.concatenate([p[0] for p in parts]).astype(np.float32); nrm = np.concatenate([p[1] for p in parts]).astype(np.float32)

    def reading():
        P = synth.se3(x=x0 + float(rng.uniform(-1.0, 6.0)), y=float(rng.normal(0, 0.4)), yaw=np.deg2rad(float(rng.normal(0, 2.0))))
        rd, rn = synth.make_scan(world, P, int(rng.integers(1500, 7000)), int(rng.integers(1, 1 << 30)), rings=16, max_range=14.0)
        rn = rn.astype(np.float32).copy()
        turn = rng.random(len(rn)) < 0.15                       # pairs a SurfaceNormalOutlierFilter must drop
        rn[turn] = rn[turn][:, [1, 2, 0]] * np.float32(-1.0)
        bad = np.linalg.norm(rn, axis=1) < 0.5
        rn[bad] = np.float32([0, 0, 1])
        return rd, rn, ref_inv @ P @ synth.se3(x=rng.normal(0, 0.05), y=rng.normal(0, 0.05), z=rng.normal(0, 0.02), yaw=np.deg2rad(rng.normal(0, 0.5)),
                                               roll=np.deg2rad(rng.normal(0, 0.3)))
    return ref, nrm, reading


t0 = time.time(); n = 0; nprob = 0; kinds = {}
while time.time() - t0 < budget:
    chain = dict(CHAIN, max_dist=float(rng.choice([0.5, 1.0, 2.0, 2.0])), trim_ratio=float(rng.choice([0.7, 0.85, 0.85, 0.95])), quantile_scale=1.0,
                 knn=int(rng.choice([1, 1, 2, 3, 4])), error_minimizer=int(rng.choice([0, 0, 1, 2, 3])), bound_max_rot=0.0, bound_max_trans=0.0,
                 normal_max_angle=0.0, outlier_max_dist=0.0, robust_fct=0, robust_tuning=1.0, robust_scale=1, robust_approx=0.0)
    if rng.random() < 0.25: chain.update(trim_ratio=0.5, quantile_scale=float(rng.choice([0.6, 1.0, 3.0])))      # MedianDistOutlierFilter
    if rng.random() < 0.2:                                                                                       # RobustOutlierFilter in the quantile filter's place
        chain.update(trim_ratio=1.0, quantile_scale=1.0, knn=1, robust_fct=int(rng.integers(1, 8)), robust_scale=int(rng.integers(0, 2)),
                     robust_approx=float(rng.choice([0.0, 0.0, 3.0])))
        chain.update(robust_tuning=float(rng.choice([0.5, 1.0, 2.5])) if chain["robust_scale"] else float(rng.choice([0.05, 0.2, 0.5])))
    if rng.random() < 0.25: chain.update(outlier_max_dist=float(rng.choice([0.1, 0.3, 1.0])))
    if rng.random() < 0.35: chain.update(normal_max_angle=float(rng.choice([0.3, 0.8, 1.5])))
    b = rng.random()
    if b < 0.2: chain.update(bound_max_rot=0.5, bound_max_trans=1.0)
    elif b < 0.35: chain.update(bound_max_rot=0.02, bound_max_trans=0.03)
    use_nrm = chain["normal_max_angle"] > 0
    DT = np.float64 if rng.random() < 0.3 else np.float32               # PointMatcher<double> in three batches of ten
    oo = o64 if DT == np.float64 else o32
    ctx = icp.Context(0, **chain)
    scenes = [scene() for _ in range(int(rng.integers(1, 4)))]
    mids = [ctx.set_map(s[0].astype(DT), s[1].astype(DT), dtype=DT) for s in scenes]
    P = int(rng.choice([1, 2, 5, 9, 16]))
    which = [int(rng.integers(0, len(scenes))) for _ in range(P)]
    rds, rns, T0s = zip(*[scenes[w][2]() for w in which])
    ok = [k for k in range(P) if len(rds[k]) >= 50]
    if not ok: ctx.close(); continue
    Ts, st = ctx.align_batch([mids[which[k]] for k in ok], [rds[k].astype(DT) for k in ok], [T0s[k] for k in ok], raise_on_error=False,
                             normals=[rns[k].astype(DT) for k in ok] if use_nrm else None, dtype=DT)
    for j, k in enumerate(ok):
        s = scenes[which[k]]
        o = oo.icp(rds[k].astype(DT), s[0].astype(DT), s[1].astype(DT), T0s[k], reading_nrm=rns[k].astype(DT) if use_nrm else None, **chain)
        try:
            assert st[j]["status"] == o["status"]
            if o["status"] == 0:
                # (welsch: exp() is a library function on both sides -- a weight that underflows to zero in one and to a denormal in the
                # other moves a pair of weight ~1e-45 between "kept" and "dropped")
                slack = 8 if chain["robust_fct"] == 2 else 0
                assert st[j]["iterations"] == o["iterations"] and st[j]["n_finite"] == o["n_finite"] and abs(st[j]["n_kept"] - o["n_kept"]) <= slack
                if DT == np.float32: assert np.float32(st[j]["trim_limit"]) == np.float32(o["trim_limit"])
                else: assert st[j]["trim_limit"] == o["trim_limit"] or abs(st[j]["trim_limit"] - o["trim_limit"]) <= 1e-9 * o["trim_limit"]
                dt, dr = pose_error(o["T"], Ts[j])
                assert dt < 1e-5 and dr < 1e-5, (dt, dr)
        except AssertionError:
            print("MISMATCH batch", n, "problem", j, "of", len(ok), "dtype", DT.__name__, "chain", chain, "device", st[j]["status"], st[j]["iterations"], st[j]["n_finite"], st[j]["n_kept"],
                  st[j]["trim_limit"], "oracle", o["status"], o["iterations"], o["n_finite"], o["n_kept"], o["trim_limit"], file=sys.stderr)
            raise
        kinds[o["status"]] = kinds.get(o["status"], 0) + 1
        nprob += 1
    ctx.close()
    n += 1
print("stress_chain: %d batches, %d ICP runs in %.0f s (by oracle status: %s), all equal to the oracle" % (n, nprob, time.time() - t0, kinds))
