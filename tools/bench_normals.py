# Surface-normal filter timing (GPU box): device-resident clouds, kernel time from the profile API,
# wall time per call (index build + kernel), CPU oracle on one core for scale.
import sys, time, json
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
from pgslam_amd import icp
from bench import build_pairs, build_workload
from oracle import Oracle

xyz, nrm, poses = build_pairs(100000)
w = build_workload(100000, 1000000, 16)
dev = torch.device('cuda', 0)
ctx = icp.Context(0)
out = {}
for name, cloud in (("keyframe_100k", xyz[0]), ("map_1M", w.map_xyz)):
    d = torch.from_numpy(np.ascontiguousarray(cloud)).to(dev)
    for knn in (10, 20):
        ctx.surface_normals(d, knn=knn, max_dist=2.0)
        ctx.profile_enable(True); ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            n = ctx.surface_normals(d, knn=knn, max_dist=2.0)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
        ctx.profile_enable(False)
        prof = ctx.profile()
        k_ms = prof["surface_normals"]["total_ms"] / prof["surface_normals"]["launches"]
        ok = np.abs(np.sum(n.cpu().numpy() * (nrm[0] if name.startswith("key") else w.map_nrm), 1))
        out[f"{name}_knn{knn}"] = dict(points=int(cloud.shape[0]), kernel_ms=k_ms, wall_ms=wall * 1e3,
                                       mpoints_per_s_kernel=cloud.shape[0] / k_ms / 1e3, median_abs_cos_to_true_normal=float(np.median(ok)))
o = Oracle(np.float32)
t0 = time.perf_counter(); o.surface_normals(xyz[0], 10, 2.0); out["cpu_oracle_1core_100k_knn10_ms"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(out, indent=1))
