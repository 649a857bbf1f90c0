"""Multi-process CPU tests of the loop-closure dispatcher (gloo, world_size 2):
sharding + padding + all-gather give every rank the single-process edge list
(SURVEY.md Appendix B.10).  No GPU: the per-pair ICP is replaced by a
deterministic stand-in, the collective path is the real one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pgslam_amd import loop_closure as lc


def fake_candidates(n):
    out = []
    for i in range(n):
        npts = 100 + 37 * (i % 5)                       # uneven costs -> uneven shards
        out.append(lc.Candidate(from_id=1000 + i, to_id=2000 + i, reading=np.zeros((npts, 3), np.float32),
                                ref_xyz=np.zeros((3 * npts, 3), np.float32), ref_nrm=np.zeros((3 * npts, 3), np.float32),
                                T_init=np.eye(4)))
    return out


def fake_align(ctx, cands, cfg):
    """stand-in for the GPU batch: results are a pure function of the pair ids"""
    edges = np.zeros(len(cands), dtype=lc.EDGE_DTYPE)
    for k, c in enumerate(cands):
        T = np.eye(4)
        T[0, 3] = 0.001 * c.from_id
        stats = dict(status=0, iterations=5 + c.to_id % 3, max_iter_reached=(c.to_id % 7 == 0), overlap=0.7 + 0.05 * (c.to_id % 6),
                     cov=np.eye(6) * (1 + c.to_id))
        edges[k] = lc.make_edge(c.from_id, c.to_id, T, stats, residual=100.0 * (c.to_id % 4), cfg=cfg)
    return edges


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = lc.LoopClosureConfig()
    edges = lc.close_loops(None, fake_candidates(n_pairs), cfg, rank=rank, world_size=world, align_fn=fake_align)
    np.save(os.path.join(path, f"edges_{rank}.npy"), edges)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [13, 64])
def test_allgather_equals_single_process(tmp_path, n_pairs):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    cfg = lc.LoopClosureConfig()
    single = lc.close_loops(None, fake_candidates(n_pairs), cfg, align_fn=fake_align)
    assert np.all(single["from_id"] == 1000 + np.arange(n_pairs))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"edges_{r}.npy"))
        assert got.tobytes() == single.tobytes()
    # acceptance follows LoopCloser::CheckIcpResult
    acc = lc.accepted_constraints(single)
    for f, t, T, cov in acc:
        assert t % 7 != 0 and 0.7 + 0.05 * (t % 6) >= 0.8 and cov.shape == (6, 6) and T.shape == (4, 4)
    assert 0 < len(acc) < n_pairs


def test_shards_are_a_balanced_partition():
    cands = fake_candidates(50)
    costs = [c.reading.shape[0] + c.ref_xyz.shape[0] for c in cands]
    parts = [lc.shard(costs, 8, r) for r in range(8)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(50))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)
    assert [len(lc.shard([1] * 512, 8, r)) for r in range(8)] == [64] * 8      # BASELINE configs[4]: 64 pairs/GPU


def test_edge_record_layout():
    e = lc.make_edge(3, 9, np.arange(16).reshape(4, 4), dict(status=0, iterations=4, max_iter_reached=False, overlap=0.9,
                                                              cov=np.arange(36).reshape(6, 6)), 12.5, lc.LoopClosureConfig())
    raw = e.tobytes()
    assert len(raw) == 512
    assert np.frombuffer(raw[:16], dtype="<i8").tolist() == [3, 9]
    assert np.frombuffer(raw[16:32], dtype="<i4").tolist() == [1, 0, 4, 0]
    assert np.frombuffer(raw[32:48], dtype="<f8").tolist() == [0.9, 12.5]
    assert np.frombuffer(raw[48:48 + 128], dtype="<f8").tolist() == list(range(16))
