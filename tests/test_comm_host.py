"""The C ABI's collective at world sizes > 1 without devices: pgicp_allgather_edges (pack -> transport -> unpack,
pgslam_amd/csrc/pgicp_comm.cpp) with the HOST transport of pgicp_comm_create_host -- the "fake all-gather" of
SURVEY.md section 4 T4 / Appendix B.10.  Every rank is a process of its own; each derives its shard and the block size
from the candidates' costs alone (pgicp_shard_pairs / pgicp_shard_slots), contributes the edges of its pairs and must
end with the byte-identical list a single process produces.  Covers uneven LPT shards, ranks without pairs, pairs
nobody reports, and several collectives on one communicator.  No GPU; no compute calls.

What the list stands for in the reference: the constraints OptimizerMT::Main drains into one solve
(/root/reference/src/pgslam/OptimizerMT.hpp:59-65, payload Optimizer.h:22)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from pgslam_amd import icp
from pgslam_amd import loop_closure as lc


def costs_of(n_pairs):
    return np.array([100_000 + 37_000 * (i % 5) + 1_000 * (i % 3) for i in range(n_pairs)], dtype=np.int64)


def edge_of(i, round_):
    """the record of pair i: a pure function of the pair index (and of the collective's round)"""
    e = np.zeros((), dtype=lc.EDGE_DTYPE)
    e["from_id"], e["to_id"] = 1000 + i, 2000 + i
    e["status"], e["iterations"], e["accepted"] = 0, 4 + (i + round_) % 9, int(i % 3 != 0)
    e["max_iter_reached"] = int(i % 7 == 0)
    e["overlap"], e["residual"] = 0.5 + 0.001 * i, 10.0 * i + round_
    T = np.eye(4)
    T[:3, 3] = [0.01 * i, -0.02 * i, 0.001 * round_]
    e["T_from_to"] = T.reshape(16)
    e["cov"] = (np.eye(6) * (1.0 + i)).reshape(36)
    return e


def rank_main(world, rank, path, n_pairs, dropped, rounds, out_dir):
    costs = costs_of(n_pairs)
    slots = icp.shard_slots(costs, world)
    comm = icp.Comm.host(world, rank, path, slots)
    assert comm.info() == (world, rank)
    mine = [i for i in icp.shard_pairs(costs, world, rank)]
    got = []
    for r in range(rounds):
        report = [i for i in mine if i not in dropped]
        local = np.zeros(len(report), dtype=lc.EDGE_DTYPE)
        for k, i in enumerate(report):
            local[k] = edge_of(i, r)
        got.append(comm.allgather_edges(local, np.asarray(report, dtype=np.int32), slots, n_pairs))
    np.save(os.path.join(out_dir, f"edges_{rank}.npy"), np.stack(got))
    comm.close()


def expected(n_pairs, dropped, rounds):
    out = np.zeros((rounds, n_pairs), dtype=lc.EDGE_DTYPE)
    for r in range(rounds):
        for i in range(n_pairs):
            if i in dropped:
                out[r, i]["from_id"] = out[r, i]["to_id"] = out[r, i]["status"] = -1
            else:
                out[r, i] = edge_of(i, r)
    return out


@pytest.mark.parametrize("world,n_pairs,dropped", [(2, 13, ()), (3, 64, (5, 40)), (8, 50, ()), (8, 5, (2,)), (1, 9, ())])
def test_host_allgather_gives_every_rank_the_single_process_list(tmp_path, world, n_pairs, dropped):
    rounds = 3
    path = str(tmp_path / "comm.shm")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=rank_main, args=(world, r, path, n_pairs, tuple(dropped), rounds, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    want = expected(n_pairs, dropped, rounds)
    # the shards really are uneven / some ranks really are empty in the cases that claim it
    sizes = [len(icp.shard_pairs(costs_of(n_pairs), world, r)) for r in range(world)]
    assert sum(sizes) == n_pairs
    if (world, n_pairs) == (8, 5):
        assert sizes.count(0) == 3
    if (world, n_pairs) == (8, 50):
        assert len(set(sizes)) > 1
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"edges_{r}.npy"))
        assert got.tobytes() == want.tobytes(), f"rank {r}"
    assert not os.path.exists(path)          # the last rank to leave removes the file


def stale_file(path, world, slots):
    """what a crashed run leaves behind: a file of the right size with the magic set, counters mid-collective and garbage
    records -- byte layout of ShmHeader in pgicp_comm.cpp (magic, world, max_slots, attached, then a 64-byte line per rank)"""
    header = (32 + 64 * world + 64 + 4095) // 4096 * 4096
    buf = np.zeros(header + 512 * world * max(1, slots), dtype=np.uint8)
    buf[header:] = 0xAB
    h = buf[:header].view(np.uint64)
    h[0], h[1], h[2], h[3] = 0x5047494350434F4D, world, max(1, slots), world
    for r in range(world):
        h[8 + 8 * r] = 7          # arrived
        h[8 + 8 * r + 1] = 7      # left
    buf.tofile(path)


@pytest.mark.parametrize("late_rank0", [False, True])
def test_host_comm_ignores_a_stale_file_of_a_crashed_run(tmp_path, late_rank0):
    """a rank must never complete a collective against the counters of a file nobody is alive behind.
    Rank 0 makes a fresh inode and answers every rank's hello; with rank 0 started LATE the others have already mapped the
    stale file and must move over to the new one."""
    import time
    world, n_pairs, rounds = 3, 20, 2
    path = str(tmp_path / "comm.shm")
    slots = icp.shard_slots(costs_of(n_pairs), world)
    stale_file(path, world, slots)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=rank_main, args=(world, r, path, n_pairs, (), rounds, str(tmp_path))) for r in range(world)]
    order = [1, 2, 0] if late_rank0 else [0, 1, 2]
    for k, r in enumerate(order):
        if late_rank0 and r == 0:
            time.sleep(1.0)
        procs[r].start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    want = expected(n_pairs, (), rounds)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"edges_{r}.npy"))
        assert got.tobytes() == want.tobytes(), f"rank {r}"
    assert not os.path.exists(path)


def test_host_comm_refuses_what_does_not_fit(tmp_path):
    comm = icp.Comm.host(1, 0, str(tmp_path / "c.shm"), 2)
    local = np.zeros(3, dtype=lc.EDGE_DTYPE)
    with pytest.raises(icp.PgicpError):
        comm.allgather_edges(local, np.arange(3, dtype=np.int32), 3, 3)           # more slots than the file holds
    with pytest.raises(icp.PgicpError):
        comm.allgather_edges(local[:1], np.array([7], dtype=np.int32), 1, 3)       # pair index out of range
    out = comm.allgather_edges(local[:0], np.zeros(0, dtype=np.int32), 0, 2)       # nothing to gather: all empty
    assert np.all(out["from_id"] == -1) and np.all(out["status"] == -1)
    comm.close()
    with pytest.raises(icp.PgicpError):
        icp.Comm.host(2, 2, str(tmp_path / "d.shm"), 1)                            # rank outside the world
