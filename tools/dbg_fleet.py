import sys
sys.path.insert(0, '.')
import numpy as np, torch
from bench import build_drive, CHAIN
from pgslam_amd import icp
from pgslam_amd.local_mapper import Keyframe, LocalMapperConfig, StreamingFleet
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cap, stride = 20, 3
first = (cap - 1) * stride
poses, odom, xyz, nrm = build_drive(first + 41, 100000, 0.35)
dev = torch.device('cuda', 0)
rebase = poses[first] @ np.linalg.inv(odom[first])
odom = [poses[s] if s < first else rebase @ odom[s] for s in range(len(odom))]
n_total = first + 12
d_xyz = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in xyz[:n_total]]
d_nrm = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in nrm[:n_total]]
ctx = icp.Context(0, **CHAIN)
f = StreamingFleet(ctx, V, LocalMapperConfig(capacity=cap, overlap_threshold=0.8, chain=dict(CHAIN)))
for m in f.mappers:
    for k in range(cap - 1):
        s = k * stride
        m.window.append(Keyframe(m.next_kf_id, d_xyz[s], d_nrm[s], odom[s].copy())); m.next_kf_id += 1
f.step([odom[first]] * V, [d_xyz[first]] * V, [d_nrm[first]] * V)
for s in range(first + 1, n_total):
    f.step([odom[s]] * V, [d_xyz[s]] * V, [d_nrm[s]] * V)
    print('scan', s, 'queue', ctx.debug_counters(), file=sys.stderr)
