"""Randomised stress of the BATCH path (GPU box): batches of 1-24 problems over 1-4 maps of random scenes (two scans, or
a scan ahead of / beside a short map), whole ICP runs against the CPU oracle -- status, iteration count, n_finite, n_kept,
threshold (bit for bit in float) and the transform to 1e-5 m / 1e-5 rad.  tools/stress_batch.py [seconds] [seed]"""
import sys, time, importlib
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from pgslam_amd import icp, synth
orc = importlib.import_module("oracle.oracle")
CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o32 = orc.Oracle(np.float32)
world = synth.make_world()


def pose_error(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    return float(np.linalg.norm(d[:3, 3])), float(np.linalg.norm([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) / 2.0)


def scene():
    if rng.random() < 0.4:
        t = synth.make_two_scans(int(rng.integers(2000, 7000)), rings=16)
        return t["ref_xyz"], t["ref_nrm"], lambda: (t["reading_xyz"][:int(rng.integers(500, len(t["reading_xyz"]) + 1))],
                                                    t["T_truth"] @ synth.se3(x=rng.normal(0, 0.1), y=rng.normal(0, 0.1), yaw=np.deg2rad(rng.normal(0, 1.0))))
    x0 = float(rng.uniform(-45, 20))
    poses = [synth.se3(x=x0 + 2.0 * k) for k in range(int(rng.integers(1, 4)))]
    ref_inv = synth.se3_inv(poses[0])
    parts = []
    for P in poses:
        x, nn = synth.make_scan(world, P, int(rng.integers(2000, 6000)), int(rng.integers(1, 1 << 30)), rings=16, max_range=14.0)
        parts.append(synth.transform_cloud(ref_inv @ P, x.astype(np.float64), nn.astype(np.float64)))
    ref = np.concatenate([p[0] for p in parts]).astype(np.float32); nrm = np.concatenate([p[1] for p in parts]).astype(np.float32)

    def reading():
        P = synth.se3(x=x0 + float(rng.uniform(-1.0, 12.0)), y=float(rng.normal(0, 0.4)), yaw=np.deg2rad(float(rng.normal(0, 2.0))))
        rd, _ = synth.make_scan(world, P, int(rng.integers(1500, 7000)), int(rng.integers(1, 1 << 30)), rings=16, max_range=14.0)
        return rd, ref_inv @ P @ synth.se3(x=rng.normal(0, 0.05), y=rng.normal(0, 0.05), yaw=np.deg2rad(rng.normal(0, 0.5)))
    return ref, nrm, reading


t0 = time.time(); n = 0; nprob = 0
while time.time() - t0 < budget:
    chain = dict(CHAIN, max_dist=float(rng.choice([0.5, 1.0, 2.0, 2.0])), trim_ratio=float(rng.choice([0.7, 0.85, 0.85, 0.95])), quantile_scale=1.0)
    ctx = icp.Context(0, **chain)
    scenes = [scene() for _ in range(int(rng.integers(1, 5)))]
    mids = [ctx.set_map(s[0], s[1]) for s in scenes]
    P = int(rng.choice([1, 2, 5, 9, 16, 24]))
    which = [int(rng.integers(0, len(scenes))) for _ in range(P)]
    rds, T0s = zip(*[scenes[w][2]() for w in which])
    ok = [k for k in range(P) if len(rds[k]) >= 50]
    if not ok: ctx.close(); continue
    Ts, st = ctx.align_batch([mids[which[k]] for k in ok], [rds[k] for k in ok], [T0s[k] for k in ok], raise_on_error=False)
    for j, k in enumerate(ok):
        s = scenes[which[k]]
        o = o32.icp(rds[k], s[0], s[1], T0s[k], **chain)
        try:
            assert st[j]["status"] == o["status"]
            if o["status"] == 0:
                assert st[j]["iterations"] == o["iterations"] and st[j]["n_finite"] == o["n_finite"] and st[j]["n_kept"] == o["n_kept"]
                assert st[j]["trim_limit"] == o["trim_limit"]
                dt, dr = pose_error(o["T"], Ts[j])
                assert dt < 1e-5 and dr < 1e-5, (dt, dr)
        except AssertionError:
            print("MISMATCH batch", n, "problem", j, "of", len(ok), "chain", chain, "device", st[j]["status"], st[j]["iterations"], st[j]["n_finite"], st[j]["n_kept"],
                  st[j]["trim_limit"], "oracle", o["status"], o["iterations"], o["n_finite"], o["n_kept"], o["trim_limit"], file=sys.stderr)
            raise
        nprob += 1
    ctx.close()
    n += 1
print("stress_batch: %d batches, %d ICP runs in %.0f s, all equal to the oracle" % (n, nprob, time.time() - t0))
