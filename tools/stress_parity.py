"""Randomised parity stress (GPU box): HIP matcher state against the CPU oracle over random scenes, poses, chain parameters
and iteration counts -- the same comparison tests/test_gpu_matcher_state.py makes (ids, squared distances, threshold, n_finite,
n_kept, bit for bit in float), many more cases.  tools/stress_parity.py [seconds] [seed]"""
import sys, time, importlib
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from pgslam_amd import icp, synth
from test_gpu_matcher_state import check_state, CHAIN
orc = importlib.import_module("oracle.oracle")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1       # run only this case of the sequence (to reproduce a mismatch)
max_cases = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
rng = np.random.default_rng(seed)
o32, o64 = orc.Oracle(np.float32), orc.Oracle(np.float64)
ctx = icp.Context(0, **CHAIN)
t0 = time.time(); n = 0
world = synth.make_world()
while time.time() - t0 < budget and n < max_cases:
    kind = rng.integers(0, 4)
    chain = dict(CHAIN, max_dist=float(rng.choice([0.3, 0.5, 1.0, 2.0, 2.0, 5.0])), trim_ratio=float(rng.choice([0.5, 0.7, 0.85, 0.85, 0.97, 1.0])))
    chain["quantile_scale"] = 1.0                    # (set_params keeps what it is not given: every case names every field it varies)
    if rng.random() < 0.15: chain["quantile_scale"] = float(rng.choice([0.4, 3.0])); chain["trim_ratio"] = 0.5
    its = tuple(sorted(set(int(x) for x in rng.integers(1, 9, size=3))))
    if kind == 0:                                   # two scans, random initial error
        t = synth.make_two_scans(int(rng.integers(3000, 9000)), rings=16)
        T0 = t["T_truth"] @ synth.se3(x=rng.normal(0, 0.3), y=rng.normal(0, 0.3), z=rng.normal(0, 0.1), yaw=np.deg2rad(rng.normal(0, 3.0)))
        rd, ref, nrm = t["reading_xyz"], t["ref_xyz"], t["ref_nrm"]
    else:                                           # a scan some metres ahead of / beside a short map
        x0 = float(rng.uniform(-45, 20))
        poses = [synth.se3(x=x0 + 2.0 * k, y=float(rng.normal(0, 0.2))) for k in range(int(rng.integers(1, 4)))]
        ref_inv = synth.se3_inv(poses[0])
        parts = []
        for k, P in enumerate(poses):
            x, nn = synth.make_scan(world, P, int(rng.integers(3000, 8000)), int(rng.integers(1, 1 << 30)), rings=16, max_range=float(rng.choice([10.0, 14.0, 25.0])))
            parts.append(synth.transform_cloud(ref_inv @ P, x.astype(np.float64), nn.astype(np.float64)))
        ref = np.concatenate([p[0] for p in parts]).astype(np.float32); nrm = np.concatenate([p[1] for p in parts]).astype(np.float32)
        ahead = float(rng.uniform(-2.0, 14.0))
        P = synth.se3(x=x0 + ahead, y=float(rng.normal(0, 0.5)), yaw=np.deg2rad(float(rng.normal(0, 3.0))))
        rd, _ = synth.make_scan(world, P, int(rng.integers(3000, 9000)), int(rng.integers(1, 1 << 30)), rings=16, max_range=14.0)
        T0 = ref_inv @ P @ synth.se3(x=rng.normal(0, 0.05), y=rng.normal(0, 0.05), yaw=np.deg2rad(rng.normal(0, 0.5)))
    if len(rd) < 100 or len(ref) < 100: continue
    dbl = rng.random() < 0.15
    if only >= 0 and n != only:
        n += 1
        if n > only: break
        continue
    try:
        if dbl: check_state(ctx, o64, rd.astype(np.float64), ref.astype(np.float64), nrm.astype(np.float64), T0, its, chain=chain, dtype=np.float64, rtol=1e-9)     # (small squared distances see the last ulp of the transforms: absolute 3e-17)
        else: check_state(ctx, o32, rd, ref, nrm, T0, its, chain=chain)
    except AssertionError as e:
        print("MISMATCH case", n, "kind", kind, "chain", chain, "its", its, "double", dbl, "sizes", len(rd), len(ref), file=sys.stderr)
        raise
    n += 1
print("stress_parity: %d random cases in %.0f s, all equal to the oracle" % (n, time.time() - t0))
