"""Batched loop-closure ICP dispatcher: shard candidate pairs over the GPUs of a
node, align each shard as one device batch, all-gather the SE(3) edges.

Reference behaviour being generalised (paths relative to
/root/reference/src/pgslam/): LoopCloserMT pops ONE vertex at a time and runs
one ICP (LoopCloserMT.hpp:45-67, LoopCloser.hpp:83-110); every accepted result
becomes an Optimizer::InputData = (from, to, T, cov) (Optimizer.h:22) and
OptimizerMT drains ALL queued constraints into one solve (OptimizerMT.hpp:59-65).
Candidate ICPs are independent of each other (LoopCloser.hpp:95-98), so here the
queue is processed as a batch: rank r takes the pairs pgicp_shard_pairs gives
it, runs them concurrently on its GPU, evaluates LoopCloser::CheckIcpResult
(LoopCloser.hpp:308-340) and the ranks exchange fixed-size 512-byte edge records
with ONE all-gather (RCCL over xGMI when the backend is "nccl"; the payload is
KBs, so the step is latency-, not bandwidth-bound).  The pose-graph solve that
consumes the gathered edges stays on the host (north_star).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import icp

EDGE_DTYPE = np.dtype([("from_id", "<i8"), ("to_id", "<i8"), ("accepted", "<i4"), ("status", "<i4"),
                       ("iterations", "<i4"), ("max_iter_reached", "<i4"), ("overlap", "<f8"), ("residual", "<f8"),
                       ("T_from_to", "<f8", (16,)), ("cov", "<f8", (36,)), ("reserved", "<f8", (6,))])
assert EDGE_DTYPE.itemsize == C.sizeof(icp.Edge) == 512


@dataclass
class Candidate:
    """One loop-closure candidate: keyframe cloud `reading` (vertex to_id) against the candidate
    local map (reference vertex from_id), initial guess T_refkf_kf (LoopCloser.hpp:95)."""
    from_id: int
    to_id: int
    reading: object            # (N,3) float32 numpy array or CUDA tensor
    ref_xyz: object
    ref_nrm: object
    T_init: np.ndarray


@dataclass
class LoopClosureConfig:
    overlap_threshold: float = 0.8            # LoopCloser.hpp:18
    residual_error_threshold: float = 5000.0  # LoopCloser.hpp:19
    chain: dict = field(default_factory=dict)


def shard(costs, world_size, rank):
    return icp.shard_pairs(np.asarray(costs, dtype=np.int64), world_size, rank)


def make_edge(from_id, to_id, T, stats, residual, cfg: LoopClosureConfig) -> np.ndarray:
    e = np.zeros((), dtype=EDGE_DTYPE)
    e["from_id"], e["to_id"] = from_id, to_id
    e["status"] = stats["status"]
    e["iterations"] = stats["iterations"]
    e["max_iter_reached"] = int(stats["max_iter_reached"])
    e["overlap"] = stats["overlap"]
    e["residual"] = residual
    e["T_from_to"] = np.asarray(T, dtype=np.float64).reshape(16)
    e["cov"] = np.asarray(stats["cov"], dtype=np.float64).reshape(36)
    e["accepted"] = int(icp.check_icp_result(stats, residual, cfg.overlap_threshold, cfg.residual_error_threshold))
    return e


def align_local(ctx: icp.Context, cands, cfg: LoopClosureConfig):
    """Run this rank's candidates as ONE device batch: ICP::operator() per pair (centred index per
    candidate map) and ComputeResidualError's chain on the result (LoopCloser.hpp:343-365), fused."""
    if cfg.chain:
        ctx.set_params(**cfg.chain)
    if not cands:
        return np.zeros(0, dtype=EDGE_DTYPE)
    map_ids = ctx.set_maps([c.ref_xyz for c in cands], [c.ref_nrm for c in cands], center=True)
    readings = [c.reading for c in cands]
    # ICP and residual check of the result in one device call (pgicp_align_residual_batch: the residual pass is seeded with
    # the last iteration's correspondences)
    Ts, sa, residual, _, _ = ctx.align_residual_batch(map_ids, readings, [c.T_init for c in cands], raw_stats=True)
    # the edge records column by column, straight from the pgicp_stats records (512 per-candidate dictionaries and record
    # assignments were a tenth of a step)
    edges = np.zeros(len(cands), dtype=EDGE_DTYPE)
    edges["from_id"] = [c.from_id for c in cands]
    edges["to_id"] = [c.to_id for c in cands]
    for k in ("status", "iterations", "max_iter_reached", "overlap", "cov"):
        edges[k] = sa[k]
    edges["residual"] = residual
    edges["T_from_to"] = np.asarray(Ts, dtype=np.float64).reshape(len(cands), 16)
    edges["accepted"] = icp.check_icp_results(sa, residual, cfg.overlap_threshold, cfg.residual_error_threshold)
    for m in map_ids:
        ctx.destroy_map(m)
    return edges


def allgather_edges_rccl(comm: "icp.Comm", local_edges: np.ndarray, pair_index, costs) -> np.ndarray:
    """The product's collective: pgicp_allgather_edges (one ncclAllGather of fixed-size blocks over RCCL / xGMI, through
    the C ABI).  `costs` are the candidates' costs the shard was made from: every rank derives the block size from them
    (icp.shard_slots), so no size is exchanged.  Pairs nobody reported come back with from_id -1."""
    slots = icp.shard_slots(costs, comm.world_size)
    return comm.allgather_edges(local_edges, pair_index, slots, len(costs))


def allgather_edges(local_edges: np.ndarray, pair_index: np.ndarray, n_pairs: int, group=None, device=None):
    """Every rank ends with the SAME list of n_pairs edges, ordered by pair index.

    Ranks may hold different numbers of pairs, so each pads its buffer to the
    largest shard with records whose from_id is -1; one all_gather of a
    contiguous uint8 tensor moves everything.  Without an initialised process
    group (single GPU) this degenerates to a local copy -- the host "fake
    collective" of SURVEY.md Appendix B.10."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    out = np.zeros(n_pairs, dtype=EDGE_DTYPE)
    out["from_id"] = out["to_id"] = out["status"] = -1          # a pair nobody reported: the C ABI's marking
    if world == 1:
        out[pair_index] = local_edges
        return out
    cap = -(-n_pairs // world)
    # shards are LPT-balanced, not exactly equal: agree on the largest one
    cnt = torch.tensor([len(local_edges)], dtype=torch.int64, device=device)
    dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=group)
    cap = max(cap, int(cnt.item()))
    rec = np.zeros(cap, dtype=[("pair", "<i8"), ("edge", EDGE_DTYPE)])
    rec["pair"] = -1
    rec["pair"][: len(local_edges)] = pair_index
    rec["edge"][: len(local_edges)] = local_edges
    send = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
    if device is not None:
        send = send.to(device)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    allrec = recv.cpu().numpy().view(rec.dtype)
    valid = allrec["pair"] >= 0
    out[allrec["pair"][valid]] = allrec["edge"][valid]
    return out


def close_loops(ctx, candidates, cfg: LoopClosureConfig, rank=0, world_size=1, group=None, device=None,
                align_fn=align_local, comm=None):
    """Shard -> align -> all-gather.  Returns the full edge list (identical on every rank).  With `comm` (icp.Comm)
    the gather is the C ABI's RCCL collective; without, torch.distributed's (the gloo tests of the host logic)."""
    costs = [int(c.reading.shape[0]) + int(c.ref_xyz.shape[0]) for c in candidates]
    mine = shard(costs, world_size, rank)
    local = align_fn(ctx, [candidates[i] for i in mine], cfg)
    if comm is not None:
        return allgather_edges_rccl(comm, local, mine, costs)
    return allgather_edges(local, mine, len(candidates), group=group, device=device)


def accepted_constraints(edges: np.ndarray):
    """What Optimizer::AddNewData receives (Optimizer.hpp:25-30): (from, to, T 4x4, COV 6x6)."""
    return [(int(e["from_id"]), int(e["to_id"]), e["T_from_to"].reshape(4, 4).copy(), e["cov"].reshape(6, 6).copy())
            for e in edges if e["from_id"] >= 0 and e["accepted"]]
