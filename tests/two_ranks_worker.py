"""One rank of tests/test_gpu_two_ranks.py: a process of its own, GPU 0, host transport (see there)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def candidates(n_pairs, n_pts):
    from pgslam_amd import synth, loop_closure as lc
    ps = synth.make_pairs(n_pairs, n_pts=n_pts, rings=16)
    return [lc.Candidate(from_id=100 + p, to_id=200 + p, reading=ps.reading_xyz[p], ref_xyz=ps.ref_xyz[p], ref_nrm=ps.ref_nrm[p], T_init=ps.T_init[p])
            for p in range(n_pairs)]


CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01, smooth_length=3, sensor_std_dev=0.01)


def main():
    world, rank, path, n_pairs, n_pts, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    from pgslam_amd import icp, loop_closure as lc
    cands = candidates(n_pairs, n_pts)
    costs = [c.reading.shape[0] + c.ref_xyz.shape[0] + 10 * (k % 3) for k, c in enumerate(cands)]      # (uneven: uneven shards)
    ctx = icp.Context(0, **CHAIN)                                            # every rank on GPU 0: the box has one
    slots = icp.shard_slots(costs, world)
    comm = icp.Comm.host(world, rank, path, slots)
    mine = lc.shard(costs, world, rank)
    local = lc.align_local(ctx, [cands[i] for i in mine], lc.LoopClosureConfig(chain=dict(CHAIN)))
    local["reserved"][:, 1] = rank + 1
    edges = comm.allgather_edges(local, np.asarray(mine, dtype=np.int32), slots, n_pairs)
    np.save(os.path.join(out_dir, f"edges_{rank}.npy"), edges)
    np.save(os.path.join(out_dir, f"mine_{rank}.npy"), np.asarray(mine))
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
