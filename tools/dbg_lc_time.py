"""Where a loop-closure step's wall time goes (host side): per device batch, the time of set_maps / align_batch /
partial_chain_batch / destroy as the host sees them (each ends with the results on the host)."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from bench import build_pairs, CHAIN
from pgslam_amd import icp, synth, loop_closure as lc
xyz, nrm, poses = build_pairs(100000)
dev = torch.device('cuda', 0)
d_xyz = [torch.from_numpy(a).to(dev) for a in xyz]; d_nrm = [torch.from_numpy(a).to(dev) for a in nrm]
nk = len(xyz); cands = []
for p in range(512):
    i = p % nk
    j = min(nk - 1, i + 1 + (p // nk) % 3) if i + 1 < nk else i - 1
    cands.append(lc.Candidate(from_id=i, to_id=j, reading=d_xyz[j], ref_xyz=d_xyz[i], ref_nrm=d_nrm[i],
                              T_init=synth.se3_inv(poses[i]) @ poses[j] @ synth.perturbation(5000 + p)))
ctx = icp.Context(0, **CHAIN)
def step(show):
    CH = int(os.environ.get('CHUNK', '128'))
    for k in range(0, 512, CH):
        cs = cands[k:k + CH]
        t = [time.perf_counter()]
        ids = ctx.set_maps([c.ref_xyz for c in cs], [c.ref_nrm for c in cs], center=True); torch.cuda.synchronize(); t.append(time.perf_counter())
        Ts, st = ctx.align_batch(ids, [c.reading for c in cs], [c.T_init for c in cs], raise_on_error=False); t.append(time.perf_counter())
        ctx.partial_chain_batch(ids, [c.reading for c in cs], Ts, raise_on_error=False); t.append(time.perf_counter())
        for m in ids: ctx.destroy_map(m)
        t.append(time.perf_counter())
        if show: print('batch', k // CH, 'set_maps %.1f align %.1f (%.1f iterations) partial %.1f destroy %.1f ms' % (
            (t[1] - t[0]) * 1e3, (t[2] - t[1]) * 1e3, np.mean([s['iterations'] for s in st]), (t[3] - t[2]) * 1e3, (t[4] - t[3]) * 1e3))
step(False)
for r in range(2):
    t0 = time.perf_counter(); step(True); print('step %.1f ms' % ((time.perf_counter() - t0) * 1e3))
