"""Round 3 study (GPU box, diagnostics build): how much lane time of the fast matcher is lost to divergence, and which
re-ordering / re-balancing scheme would recover how much of it.

The stats build (-DPGICP_KNN_STATS) dumps, for every query and pass, the candidates it evaluated in the three loops of
k_knn_grid (own row, flat walk over the neighbour rows, ring continuation) and the seed distance the pass started with.
A wave executes max-over-lanes trips of each loop; from the dump this script computes, per pass,

  util            sum(cnt) / (64 * sum over waves of the wave's maximum)        -- per loop and for the three together
  by_prev_d2      the same after a stable re-order of every problem's queries by log2-class of the seed distance
  by_prev_cnt     ... by log2-class of the candidates the previous pass evaluated (an oracle for "work is predictable")
  recycle(B, G)   blocks of B consecutive queries; every lane works G candidates per round, unfinished queries are
                  compacted into full waves between rounds: cost = sum over rounds of ceil(unfinished / 64) * G
  ideal           sum(cnt) / 64

Usage (on the GPU box): STATS=-DPGICP_KNN_STATS tools/stats_build.sh && python3 tools/studies/balance_study.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_workload, CHAIN          # noqa: E402
from pgslam_amd import icp                        # noqa: E402

B, PASSES, N = 16, 8, 100_000
w = build_workload(N, 1_000_000, 64)
dev = torch.device("cuda", 0)
rd = [torch.from_numpy(s).to(dev) for s in w.scans_xyz[:B]]
ctx = icp.Context(0, **dict(CHAIN, max_iters=PASSES, min_diff_rot=0.0, min_diff_trans=0.0))
mid = ctx.set_map(torch.from_numpy(w.map_xyz).to(dev), torch.from_numpy(w.map_nrm).to(dev))
lib = ctx.lib
total = B * N
lib.pgicp_debug_dump_setup.argtypes = [C.c_longlong, C.c_int]
lib.pgicp_debug_dump_read.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
assert lib.pgicp_debug_dump_setup(total, PASSES) == 0, "not a -DPGICP_KNN_STATS build"
ctx.align_batch(mid, rd, w.T_init[:B])
cnt = np.zeros(PASSES * total, dtype=np.uint32)
d2 = np.zeros(PASSES * total, dtype=np.float32)
assert lib.pgicp_debug_dump_read(cnt.ctypes.data, d2.ctypes.data, PASSES * total) == 0
cnt = cnt.reshape(PASSES, B, N)
d2 = d2.reshape(PASSES, B, N)
os.makedirs("gpurun_out/r3", exist_ok=True)
np.savez_compressed("gpurun_out/r3/balance_dump.npz", cnt=cnt[:, :6], d2=d2[:, :6].astype(np.float16))      # for offline what-ifs
own, flat, ring = (cnt & 1023).astype(np.int64), ((cnt >> 10) & 1023).astype(np.int64), (cnt >> 20).astype(np.int64)


def wave_cost(c):
    """c: (B, N) per-lane trips -> sum over waves of the maximum (waves = 64 consecutive queries of a problem)"""
    pad = (-c.shape[1]) % 64
    cc = np.pad(c, ((0, 0), (0, pad)))
    return int(cc.reshape(c.shape[0], -1, 64).max(axis=2).sum())


def reorder(key):
    """stable order of every problem's queries by key (B, N)"""
    return np.argsort(key, axis=1, kind="stable")


def recycle_cost(c, block, g):
    """blocks of `block` consecutive queries, rounds of g candidates, compaction between rounds"""
    pad = (-c.shape[1]) % block
    cc = np.pad(c, ((0, 0), (0, pad))).reshape(-1, block)
    cost = 0
    r = 0
    while True:
        active = (cc > r * g).sum(axis=1)
        if not active.any():
            break
        cost += int(np.ceil(active / 64.0).sum()) * g
        r += 1
    return cost


out = []
for p in range(PASSES):
    tot = own[p] + flat[p] + ring[p]
    row = dict(pass_=p, mean_own=float(own[p].mean()), mean_flat=float(flat[p].mean()), mean_ring=float(ring[p].mean()))
    ideal = tot.sum() / 64.0
    now = wave_cost(own[p]) + wave_cost(flat[p]) + wave_cost(ring[p])
    row["util_now"] = ideal / now
    row["util_own"] = own[p].sum() / 64.0 / max(1, wave_cost(own[p]))
    row["util_flat"] = flat[p].sum() / 64.0 / max(1, wave_cost(flat[p]))
    row["util_ring"] = ring[p].sum() / 64.0 / max(1, wave_cost(ring[p]))
    row["one_walk_util"] = ideal / wave_cost(tot)                     # one merged loop instead of three
    if p > 0:
        k1 = np.where(d2[p] >= 0, np.floor(np.log2(np.maximum(d2[p], 1e-12)) * 1.0), -100).astype(np.int64)
        o = reorder(k1)
        g = lambda a: np.take_along_axis(a, o, axis=1)
        row["util_by_prev_d2"] = ideal / (wave_cost(g(own[p])) + wave_cost(g(flat[p])) + wave_cost(g(ring[p])))
        prev = own[p - 1] + flat[p - 1] + ring[p - 1]
        k2 = np.floor(np.log2(np.maximum(prev, 1))).astype(np.int64)
        o = reorder(k2)
        row["util_by_prev_cnt"] = ideal / (wave_cost(g(own[p])) + wave_cost(g(flat[p])) + wave_cost(g(ring[p])))
        o = reorder(np.floor(np.log2(np.maximum(tot, 1))).astype(np.int64))          # perfect knowledge, log classes
        row["util_by_own_cnt"] = ideal / (wave_cost(g(own[p])) + wave_cost(g(flat[p])) + wave_cost(g(ring[p])))
    for blk in (256, 1024):
        for gg in (8, 16):
            row[f"recycle_{blk}_{gg}"] = ideal / recycle_cost(tot, blk, gg)
    # the tail: share of the candidates carried by the busiest 5 % of the queries
    srt = np.sort(tot.reshape(-1))
    row["top5pct_share"] = float(srt[int(0.95 * srt.size):].sum() / max(1, srt.sum()))
    row["p50"], row["p90"], row["p99"], row["max"] = (int(np.percentile(srt, q)) for q in (50, 90, 99, 100))
    out.append(row)
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in row.items()}))
os.makedirs("gpurun_out/r3", exist_ok=True)
json.dump(out, open("gpurun_out/r3/balance_study.json", "w"), indent=1)
