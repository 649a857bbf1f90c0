# streaming workload diagnostics (stats build): queue sizes / slow-path entries per scan
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from bench import build_drive, CHAIN
from pgslam_amd import icp
from pgslam_amd.local_mapper import Keyframe, LocalMapperConfig, StreamingLocalMapper
cap, stride, nscan = 20, 3, int(os.environ.get("NSCAN", "8"))
n_total = (cap - 1) * stride + nscan
poses, odom, xyz, nrm = build_drive((cap - 1) * stride + 41, 100000, 0.35)
dev = torch.device('cuda', 0)
first = (cap - 1) * stride
rebase = poses[first] @ np.linalg.inv(odom[first])
odom = [poses[s] if s < first else rebase @ odom[s] for s in range(len(odom))]
d_xyz = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in xyz[:n_total]]
d_nrm = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in nrm[:n_total]]
ctx = icp.Context(0, **CHAIN)
m = StreamingLocalMapper(ctx, LocalMapperConfig(capacity=cap, overlap_threshold=0.8, chain=dict(CHAIN)))
for k in range(cap - 1):
    s = k * stride
    m.window.append(Keyframe(m.next_kf_id, d_xyz[s], d_nrm[s], odom[s].copy())); m.next_kf_id += 1
m.process(odom[first], d_xyz[first], d_nrm[first])
for s in range(first + 1, n_total):
    m.process(odom[s], d_xyz[s], d_nrm[s])
    st = m.last_stats
    print('scan', s, 'iters', st['iterations'], 'n_finite', st['n_finite'], 'limit %.4f' % st['trim_limit'], 'last-iteration queue', ctx.debug_counters(), file=sys.stderr)
