"""The N > 1 path on hardware, as far as a one-GPU box allows: TWO rank processes (both on GPU 0, each with its own context and
stream) run the loop-closure dispatcher end to end -- the deterministic LPT shard (pgicp_shard_pairs), the device batch of the
rank's pairs (index build + ICP + residual check), pgicp_allgather_edges over the HOST transport (pgicp_comm_create_host: RCCL
refuses two ranks on one device) -- and every rank must end with the edge list a single process makes of all pairs, bit for bit.
What the shard / gather replaces in the reference: LoopCloserMT's one-at-a-time loop and OptimizerMT's drained queue
(/root/reference/src/pgslam/LoopCloserMT.hpp:45-67, OptimizerMT.hpp:59-65).  RCCL itself is exercised at world size 1 in
tests/test_gpu_parity.py and at N > 1 by the driver's multi-GPU tier."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_rank_processes_share_the_gpu_and_gather_the_single_process_list(tmp_path):
    sys.path.insert(0, HERE)
    import two_ranks_worker as w
    from pgslam_amd import icp, loop_closure as lc
    world, n_pairs, n_pts = 2, 7, 8000
    shm = "/dev/shm/pgicp_two_ranks_%d" % os.getpid()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_ranks_worker.py"), str(world), str(r), shm, str(n_pairs), str(n_pts), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    # the single-process list: all pairs as one device batch
    cands = w.candidates(n_pairs, n_pts)
    ctx = icp.Context(0, **w.CHAIN)
    single = lc.align_local(ctx, cands, lc.LoopClosureConfig(chain=dict(w.CHAIN)))
    ctx.close()
    assert np.all(single["status"] == 0) and single["accepted"].sum() >= 1
    seen = set()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"edges_{r}.npy"))
        mine = np.load(os.path.join(str(tmp_path), f"mine_{r}.npy"))
        assert 0 < len(mine) < n_pairs
        seen |= set(int(i) for i in mine)
        who = got["reserved"][:, 1].copy()
        got["reserved"][:, 1] = 0
        assert got.tobytes() == single.tobytes(), "rank %d" % r
        assert set(np.nonzero(who == r + 1)[0]) == set(int(i) for i in mine)       # every pair was aligned by the rank the shard names
    assert seen == set(range(n_pairs))
