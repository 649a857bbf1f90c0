#!/usr/bin/env python3
"""bench.py -- ICP-converged scans/sec at a 100k-pt scan vs a 1M-pt local map.

Workload = BASELINE.json configs[1]: "Scan-to-map ICP: 100k-pt Velodyne-style
scan vs 1M-pt local map, 30 iters, 1xMI355X" (synthetic).
A "step" is one batch of `--batch` independent scan-to-map ICPs (distinct query
scans, each with its own perturbed initial guess) against the resident map;
`value` = scans whose Differential checker stopped the loop (converged, status
OK) per second, whole job.  Inputs are in HBM before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank holds a
full copy of the map and aligns its own batch -- independent replicas, weak
scaling, no data-path collective (scan-to-map ICPs of one trajectory are
sequential).  The barrier and the max-over-ranks timing use
torch.distributed (RCCL).

The JSON line also carries
  roofline     kNN kernel (dominant): algorithmic bytes per launch
               ((20 N + 12 M) per active problem) / the kernel's
               mean duration measured with HIP events on the context's stream,
               against the 8 TB/s HBM peak.
  cpu_baseline the CPU oracle (kd-tree restatement of the reference chain, kind
               "port") timed on this box's host cores on a bounded sample of the
               same scans.  Baseline, not target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md)


LINE_MAX = 6000             # the driver keeps ~8 KB of stdout: the final line stays well inside that
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config")
LINE_FACTS = ("dry_run", "scans_total", "scans_converged", "mean_iterations", "median_translation_error_m", "set_map_ms", "pairs_ok",
              "pairs_accepted", "rccl_ranks_seen", "comm_world_size", "edges_transport", "pairs_per_s_one_gpu_same_run", "speedup_vs_one_gpu",
              "speedup_vs_cpu_baseline", "launched_by", "command_wall_s")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches",
                 "active_problems_per_launch", "algorithmic_bytes_per_launch", "bound_measured", "traffic_measured_in_this_run",
                 "valu_lane_ops_per_active_query_iteration", "frac_unseeded_launches", "frac_seeded_launches", "frac_shared_map")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "single_core_scans_per_s", "mean_iterations")


def _short(v):
    """Floats of the printed line at 6 significant digits (the full record keeps every bit)."""
    if isinstance(v, float):
        return float(f"{v:.6g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _short(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x) for x in v]
    return v


def compact_line(full: dict) -> dict:
    """The line the driver parses: the contract's keys, `roofline` and `cpu_baseline` as flat objects of numbers and names (no
    per-launch arrays, no prose beyond `sample`), a few scalar facts, `legs` LAST.  Everything else lives in the full record
    (`bench_full.json`, see emit())."""
    out = {k: full[k] for k in LINE_KEYS if k in full}
    cfg = dict(out.get("config") or {})
    cfg.pop("chain", None)                              # (the chain is BASELINE's: in the full record and in DESIGN.md)
    out["config"] = cfg
    r = full.get("roofline")
    out["roofline"] = {k: r[k] for k in ROOFLINE_KEYS if k in r} if r else None
    c = full.get("cpu_baseline")
    out["cpu_baseline"] = {k: c[k] for k in CPU_KEYS if k in c} if c else None
    for k in LINE_FACTS:
        if full.get(k) is not None:
            out[k] = full[k]
    out["full_record"] = FULL_RECORD_NAME
    out["legs"] = full.get("legs")
    out = _short(out)
    # a guard, not a plan: shed the optional facts (never the contract, roofline, cpu_baseline or legs) if a line ever grows
    for k in reversed(LINE_FACTS):
        if len(json.dumps(out)) < LINE_MAX:
            break
        out.pop(k, None)
    if len(json.dumps(out)) >= LINE_MAX and out.get("cpu_baseline"):
        out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:160]
    return out


FULL_RECORD_NAME = "bench_full.json"


def write_full_record(full: dict):
    """The whole record (every leg's long form, per-launch counters, per-step timings) as a FILE next to bench.py and, on a
    gpurun box, under gpurun_out/ so that it travels back."""
    text = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, FULL_RECORD_NAME), "w") as f:
                    f.write(text + "\n")
            except OSError:
                pass


def emit(full: dict):
    """The ONE JSON line, as the last thing on stdout, short enough for the driver to keep whole (compact_line); the full
    record goes to bench_full.json.  Native libraries (RCCL prints a version banner when a communicator is made) write
    through C stdio, whose buffer would otherwise be flushed after Python's at exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    write_full_record(full)
    line = json.dumps(compact_line(full), allow_nan=False)
    assert len(line) < LINE_MAX, len(line)
    sys.stdout.flush()
    print(line, flush=True)


_DIST = {"on": False}


def ranks():
    """(world size, rank, local rank) of this process: torch.distributed.run's variables, or the ones launch_ranks set"""
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def dist_begin(backend, local_rank):
    """One process group per process, made by whichever leg needs it first ("nccl" IS RCCL on ROCm; "gloo" where only host
    numbers travel).  Returns whether the job is distributed."""
    world = ranks()[0]
    if world <= 1:
        return False
    if not _DIST["on"]:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        _DIST["on"] = True
    return True


def dist_end():
    if _DIST["on"]:
        import torch.distributed as dist
        dist.destroy_process_group()
        _DIST["on"] = False


def require_device(local_rank):
    """Fail loudly, before any rendezvous, when this rank has no GPU: the product has no CPU path (PGICP_ERR_NO_DEVICE)."""
    from pgslam_amd import icp
    icp.Context(local_rank).close()


def launch_ranks(n):
    """`bench.py --gpus N` started by hand: the parent (which never touches a GPU) starts N fresh rank processes of this
    same command line -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them --
    passes rank 0's output through and exits with the worst exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PGSLAM_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, failed_at = 0, None
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p_ in list(live):
            c = p_.poll()
            if c is None:
                continue
            live.remove(p_)
            if c != 0:
                rc = rc or c
                failed_at = failed_at or time.time()
        if failed_at and live and time.time() - failed_at > 20.0:
            for p_ in live:                      # a rank died: the others would wait at the rendezvous for ever
                p_.terminate()
            failed_at = time.time() + 1e9
    if rc:
        print(f"bench.py: a rank failed (exit code {rc}); no result line", file=sys.stderr)
    sys.exit(rc)


def _gen_scan(args):
    from pgslam_amd import synth
    kind, i, n_pts, rings, pose = args
    world = synth.make_world()
    return synth.make_scan(world, pose, n_pts, i, rings=rings)


def build_workload(n_scan, n_map, n_queries, cache_dir="/tmp"):
    """Same construction as synth.make_scan_to_map, but with the ray casting of
    the individual scans spread over host cores, and cached on disk."""
    from pgslam_amd import synth
    key = f"pgslam_amd_s2m_{n_scan}_{n_map}_{n_queries}.npz"
    path = os.path.join(cache_dir, key)
    if os.path.exists(path):
        z = np.load(path)
        return synth.ScanToMap(z["map_xyz"], z["map_nrm"], list(z["scans_xyz"]), [None] * n_queries,
                               list(z["T_truth"]), list(z["T_init"]))
    import multiprocessing as mp
    import math
    n_map_poses = 12 if n_map <= 1_500_000 else 24
    rings = 64 if n_scan >= 50_000 else 16
    jitter = np.deg2rad(synth.uniform(synth.WORLD_SEED + 17, n_map_poses, -2.0, 2.0))
    poses = [synth.se3(x=1.5 * i, yaw=float(jitter[i])) for i in range(n_map_poses)]
    T_ref, T_ref_inv = poses[-1], synth.se3_inv(poses[-1])
    per_scan = max(n_scan, -(-n_map // n_map_poses))
    order = [n_map_poses - 1] + list(range(n_map_poses - 2, -1, -1))
    qu = synth.uniform01(synth.WORLD_SEED + 33, 3 * n_queries).reshape(n_queries, 3)
    T_q = []
    for b in range(n_queries):
        dx = 0.75 + (-0.5 + 1.0 * qu[b, 0])
        dy = -0.3 + 0.6 * qu[b, 1]
        dyaw = math.radians(-3.0 + 6.0 * qu[b, 2])
        T_q.append(T_ref @ synth.se3(x=dx, y=dy, yaw=dyaw))
    jobs = [("map", 1000 + i, per_scan, rings, poses[i]) for i in order] + \
           [("query", b, n_scan, rings, T_q[b]) for b in range(n_queries)]
    nproc = max(1, min(len(jobs), (os.cpu_count() or 1), 32))
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(_gen_scan, jobs)
    parts_p, parts_n = [], []
    for k, i in enumerate(order):
        p, n = synth.transform_cloud(T_ref_inv @ poses[i], res[k][0].astype(np.float64), res[k][1].astype(np.float64))
        parts_p.append(p)
        parts_n.append(n)
    mp_, mn_ = np.concatenate(parts_p), np.concatenate(parts_n)
    sel = (np.arange(n_map, dtype=np.int64) * mp_.shape[0]) // n_map
    map_xyz, map_nrm = mp_[sel].astype(np.float32), mn_[sel].astype(np.float32)
    scans = [res[len(order) + b][0] for b in range(n_queries)]
    truth = [T_ref_inv @ T_q[b] for b in range(n_queries)]
    init = [truth[b] @ synth.perturbation(b) for b in range(n_queries)]
    try:
        np.savez(path + ".tmp.npz", map_xyz=map_xyz, map_nrm=map_nrm, scans_xyz=np.stack(scans),
                 T_truth=np.stack(truth), T_init=np.stack(init))
        os.replace(path + ".tmp.npz", path)
    except OSError:
        pass
    return synth.ScanToMap(map_xyz, map_nrm, scans, [None] * n_queries, truth, init)


def cpu_baseline(w, sample_scans, n_threads, reps=3):
    """Time the oracle (kd-tree port of the reference chain) on ALL host cores: one independent ICP
    per thread (pgslam gives each ICP object one thread, LocalizerMT.hpp:43-48), scans handed out from
    a shared counter, `reps` repetitions, median.  The kd-tree build is timed
    separately (setMap).  The single-core figure is the median of `reps` runs of one scan."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    import itertools
    import statistics
    o = Oracle(np.float32)
    t0 = time.perf_counter()
    m = o.map_create(w.map_xyz, w.map_nrm, center=True, use_kdtree=True)
    t_build = time.perf_counter() - t0
    n_thr = max(1, n_threads)
    n_scans = max(sample_scans, n_thr)                 # every core gets at least one scan

    t_one, r0 = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        r0 = o.icp_map(m, w.scans_xyz[0], w.T_init[0], **CHAIN)
        t_one.append(time.perf_counter() - t0)

    rates, results = [], None
    for _ in range(reps):
        results = [None] * n_scans
        ticket = itertools.count()                     # next() on it is atomic under the GIL

        def work():
            while True:
                b = next(ticket)
                if b >= n_scans:
                    return
                q = b % len(w.scans_xyz)
                results[b] = o.icp_map(m, w.scans_xyz[q], w.T_init[q], **CHAIN)   # ctypes call: the GIL is released

        th = [threading.Thread(target=work) for _ in range(n_thr)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        rates.append(sum(1 for r in results if r["status"] == 0 and r["converged"]) / dt)
    o.map_free(m)
    iters = float(np.mean([r["iterations"] for r in results]))
    # an installed libpointmatcher would be timed here instead (kind "reference") -- probed, never stubbed
    import ref_probe
    pr = ref_probe.probe()
    ref_line = None
    if pr["exe"]:
        try:
            rr = ref_probe.run(pr["exe"], w.scans_xyz[0], w.map_xyz, w.map_nrm, w.T_init[0], repetitions=2)
            dT = np.linalg.inv(rr["T"]) @ r0["T"]
            ref_line = dict(single_core_scans_per_s=1.0 / rr["seconds"], translation_vs_port_m=float(np.linalg.norm(dT[:3, 3])),
                            note="ICP::operator() of the installed libpointmatcher (index build included in every call)")
        except Exception as e:                  # a reference that does not run is reported, not hidden
            ref_line = dict(error=f"{type(e).__name__}: {e}")
    probe_line = dict(libpointmatcher_found=pr["found"], missing=pr["missing"][:4], build_error=pr["build_error"], reference=ref_line)
    return dict(value=statistics.median(rates), unit="scans/s", cores=n_thr, kind="port", reference_probe=probe_line,
                sample=f"{n_scans} of the benchmark's 100k-pt scans vs the 1M-pt map, kd-tree oracle (CPU restatement of the "
                       f"reference chain, not libpointmatcher), one ICP per thread on all {n_thr} host cores "
                       f"(os.cpu_count() = {os.cpu_count()}), median of {reps} repetitions; index build excluded",
                host_cores=os.cpu_count(), repetitions=reps, all_core_rates=rates,
                single_core_scans_per_s=(1.0 / statistics.median(t_one) if r0["status"] == 0 else 0.0),
                mean_iterations=iters, index_build_s=t_build)


def check_all_ranks_reported(seen, comm_world, world):
    """A multi-rank line is only a measurement if every rank's edges arrived through the communicator: the gathered records
    name the rank that aligned them (`reserved[1]`), so a rank whose shard went missing -- or a communicator smaller than the
    job -- ends the run with a non-zero exit code instead of a line (every rank holds the same gathered edges and fails alike)."""
    if len(seen) != world or comm_world != world or list(seen) != list(range(world)):
        print(f"bench.py: {world} rank(s) launched, communicator of {comm_world}, edges from ranks {list(seen)}: "
              f"not a {world}-GPU measurement; no result line", file=sys.stderr)
        sys.exit(3)


def build_pairs(n_pts, n_keyframes=24, cache_dir="/tmp"):
    """Keyframe clouds for BASELINE configs[4] (cached; generated on host cores)."""
    from pgslam_amd import synth
    path = os.path.join(cache_dir, f"pgslam_amd_kf_{n_pts}_{n_keyframes}.npz")
    jitter = np.deg2rad(synth.uniform(synth.WORLD_SEED + 51, n_keyframes, -3.0, 3.0))
    poses = [synth.se3(x=-12.0 + 1.0 * i, yaw=float(jitter[i])) for i in range(n_keyframes)]
    if os.path.exists(path):
        z = np.load(path)
        return list(z["xyz"]), list(z["nrm"]), poses
    import multiprocessing as mp
    rings = 64 if n_pts >= 50_000 else 16
    jobs = [("kf", 2000 + i, n_pts, rings, poses[i]) for i in range(n_keyframes)]
    with mp.get_context("fork").Pool(max(1, min(len(jobs), os.cpu_count() or 1, 32))) as pool:
        res = pool.map(_gen_scan, jobs)
    xyz, nrm = [r[0] for r in res], [r[1] for r in res]
    try:
        np.savez(path + ".tmp.npz", xyz=np.stack(xyz), nrm=np.stack(nrm))
        os.replace(path + ".tmp.npz", path)
    except OSError:
        pass
    return xyz, nrm, poses


def main_loopclosure_dry(args):
    """HARNESS TEST ONLY (PGSLAM_BENCH_DRY_RANKS=1, tests/test_bench_ranks.py): the N > 1 skeleton of main_loopclosure on a box
    without devices -- launch_ranks' processes, the LPT shard, pgicp_allgather_edges over the library's HOST transport, the
    every-rank-reported check, rank 0 doing all pairs alone, the line's N > 1 keys -- with the device batch replaced by a
    stand-in that computes nothing.  The line says so (`dry_run`, `data`) and its `value` measures nothing."""
    import torch.distributed as dist
    from pgslam_amd import icp, loop_closure as lc
    world, rank, _ = ranks()
    distributed = dist_begin("gloo", 0)
    n = args.pairs
    costs = [200_000 + 1_000 * (i % 7) for i in range(n)]
    mine = lc.shard(costs, world, rank)
    slots = icp.shard_slots(costs, world)
    shm = "/dev/shm/pgslam_bench_dry_%s" % os.environ.get("MASTER_PORT", "0")
    comm = icp.Comm.host(world, rank, shm, slots)
    drop = os.environ.get("PGSLAM_BENCH_DRY_DROP_RANK")

    def align_shard(idx):
        e = np.zeros(len(idx), dtype=lc.EDGE_DTYPE)
        e["from_id"], e["to_id"], e["iterations"], e["accepted"] = [1000 + i for i in idx], [2000 + i for i in idx], 5, 1
        time.sleep(2e-4 * len(idx))
        return e

    def step():
        local = align_shard(mine)
        local["reserved"][:, 1] = rank + 1
        if drop is not None and rank == int(drop):
            return comm.allgather_edges(local[:0], np.zeros(0, np.int32), slots, n)
        return comm.allgather_edges(local, np.asarray(mine, dtype=np.int32), slots, n)

    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        edges = step()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = sorted(set(int(v) - 1 for v in np.asarray(edges["reserved"])[:, 1] if v > 0))
    comm_world, _ = comm.info()
    check_all_ranks_reported(seen, comm_world, world)
    single = None
    if distributed:
        if rank == 0:
            t1 = time.perf_counter()
            for _ in range(args.steps):
                align_shard(list(range(n)))
            single = args.steps * n / (time.perf_counter() - t1)
        dist.barrier()
    comm.close()
    if rank == 0:
        value = args.steps * n / elapsed
        emit({"metric": "DRY RUN of the loop-closure rank skeleton (no device, stand-in aligner): not a measurement", "value": value,
              "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "none", "data": "dry run: nothing computed",
              "config": {"workload": f"harness test, {n} stand-in pairs", "parallelism": f"{world} rank process(es), host transport"},
              "dry_run": True, "roofline": None, "cpu_baseline": None,
              "rccl_ranks_seen": len(seen), "comm_world_size": comm_world, "pairs_per_s_one_gpu_same_run": single,
              "speedup_vs_one_gpu": value / single if single else None,
              "launched_by": "bench.py itself (launch_ranks)" if os.environ.get("PGSLAM_BENCH_SELF_LAUNCHED") else "a launcher",
              "legs": {"dry_run": True}})
    dist_end()


def main_loopclosure(args, collect=False):
    """BASELINE configs[4]: `--pairs` candidate pairs sharded over the ranks (strong scaling),
    every rank aligns its shard in device batches, one all-gather of the 512-byte edge records.
    At N > 1 rank 0 then aligns ALL pairs alone in the same run: `speedup_vs_one_gpu` is north_star's ratio."""
    if os.environ.get("PGSLAM_BENCH_DRY_RANKS") == "1" and not collect:
        return main_loopclosure_dry(args)
    import torch
    import torch.distributed as dist
    from pgslam_amd import icp, synth, loop_closure as lc

    world, rank, local_rank = ranks()
    require_device(local_rank)
    distributed = dist_begin("nccl", local_rank)
    dev = torch.device("cuda", local_rank)
    if rank == 0:
        xyz, nrm, poses = build_pairs(args.n_scan)
    if distributed:
        dist.barrier()
    if rank != 0:
        xyz, nrm, poses = build_pairs(args.n_scan)
    d_xyz = [torch.from_numpy(a).to(dev) for a in xyz]
    d_nrm = [torch.from_numpy(a).to(dev) for a in nrm]
    nk = len(xyz)
    cands = []
    for p in range(args.pairs):
        i = p % nk
        j = min(nk - 1, i + 1 + (p // nk) % 3) if i + 1 < nk else i - 1
        T_true = synth.se3_inv(poses[i]) @ poses[j]
        cands.append(lc.Candidate(from_id=i, to_id=j, reading=d_xyz[j], ref_xyz=d_xyz[i], ref_nrm=d_nrm[i],
                                  T_init=T_true @ synth.perturbation(5000 + p)))
    cfg = lc.LoopClosureConfig(chain=dict(CHAIN))
    ctx = icp.Context(local_rank, **CHAIN, check_every=args.check_every)
    costs = [c.reading.shape[0] + c.ref_xyz.shape[0] for c in cands]
    mine = lc.shard(costs, world, rank)
    # the collective of the path goes through the C ABI (pgicp_allgather_edges: one ncclAllGather over RCCL / xGMI), at
    # world size 1 too; torch.distributed only carries the communicator's unique id, the barrier and the timing
    uid = [icp.comm_unique_id() if rank == 0 else None]
    if distributed:
        dist.broadcast_object_list(uid, src=0)
    # (N > 1 has never run on hardware here: if the C ABI's communicator cannot be made on some rank -- all ranks agree on that through
    #  torch.distributed -- the edge lists travel by torch.distributed's all_gather instead, RCCL as well, and the line says so in
    #  `edges_transport`; a leg that dies takes the job's line with it)
    comm, comm_error = None, None
    try:
        comm = icp.Comm(ctx, world, rank, uid[0])
    except Exception as e:
        comm_error = f"{type(e).__name__}: {e}"
        if not distributed:
            raise
    if distributed:
        ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
            comm = None
            errs = [None] * world
            dist.all_gather_object(errs, comm_error)
            comm_error = next((e for e in errs if e), "a rank could not make the communicator")

    # --lc-contexts C > 1: the shard as C sub-batches on C contexts (C HIP streams, C host threads): one sub-batch's index build
    # and nearly empty last iterations overlap another's full launches (pgicp contexts are fully concurrent, include/pgicp.h)
    extra_ctx = [icp.Context(local_rank, **CHAIN, check_every=args.check_every) for _ in range(max(0, args.lc_contexts - 1))]

    def align_shard(idx):
        if extra_ctx and len(idx) >= 2 * len(extra_ctx) + 2:
            cs = [ctx] + extra_ctx
            C_ = len(cs)
            idx = list(idx)
            parts = [None] * C_

            def work(k):
                sub = idx[k::C_]
                parts[k] = lc.align_local(cs[k], [cands[i] for i in sub], cfg)
            th = [threading.Thread(target=work, args=(k,)) for k in range(C_)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            out_ = np.zeros(len(idx), dtype=lc.EDGE_DTYPE)
            for k in range(C_):
                out_[k::C_] = parts[k]
            return out_
        parts = [lc.align_local(ctx, [cands[i] for i in idx[k:k + args.pair_chunk]], cfg)
                 for k in range(0, len(idx), args.pair_chunk)]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=lc.EDGE_DTYPE)

    def step():
        local = align_shard(mine)
        local["reserved"][:, 1] = rank + 1          # which rank aligned the pair: the line proves the ranks it saw
        if comm is None:
            return lc.allgather_edges(local, np.asarray(mine, dtype=np.int64), len(cands), device=dev)
        return lc.allgather_edges_rccl(comm, local, mine, costs)

    for _ in range(args.warmup):
        step()
    import gc
    gc.collect()
    gc.freeze()                     # (see main(): collections inside the timed steps must not walk torch's and numpy's objects)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        edges = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- roofline of the dominant kernel (the fast matcher again), per SURVEY.md 8(d) "Config 5 bytes": per pair and
    #      iteration 20 N + 12 M, with N = M = the clouds' size and every pair holding its own map; measured with HIP
    #      events over a replay of the timed steps ----
    roofline = None
    if not args.no_profile:
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(args.steps):
            step()
        ctx.profile_enable(False)
        k = ctx.profile()["knn_grid"]
        if k["launches"]:
            alg = 20.0 * k["units"] + 12.0 * args.n_scan * k["problems"]
            avg_s = k["total_ms"] * 1e-3 / k["launches"]
            ach = alg / k["launches"] / avg_s / 1e9
            roofline = dict(bound="hbm", kernel="knn_grid", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                            avg_launch_us=avg_s * 1e6, launches=k["launches"], profiled_steps=args.steps,
                            algorithmic_bytes_per_launch=alg / k["launches"], active_problems_per_launch=k["problems"] / k["launches"],
                            **recorded_traffic("knn_traffic_loopclosure", n_scan=args.n_scan, batch=args.pairs))
    # ---- strong-scaling reference, same run: rank 0 aligns ALL pairs alone while the others wait
    single = None
    if distributed:
        every = list(range(len(cands)))
        if rank == 0:
            align_shard(every)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                align_shard(every)
            torch.cuda.synchronize()
            single = args.steps * len(cands) / (time.perf_counter() - t1)
        dist.barrier()
    # ---- ONE-GPU PROXY of the multi-GPU target (north_star: >= 3.5x at 8 GPUs on this workload; no 8-GPU node is available to
    #      this run).  The path shards with no data exchange before the final all-gather, so rank r of a
    #      W-rank job does exactly what this GPU does when handed rank r's LPT shard: the per-batch fixed costs (the unseeded
    #      first matcher launch over a smaller batch, the convergence tail, the index build at P = 512 / W maps) are all in the
    #      shard's time.  T(512) and every rank's shard are timed here back to back on the same box;
    #      predicted_speedup(W) = T(512) / (max_r T(shard r of W) + t_allgather), t_allgather = the world-1 collective's
    #      latency (pack + ncclAllGather + unpack of the same 512 records; over xGMI the 256 KiB are latency-bound too).
    #      What the proxy cannot see: 8 host processes sharing the box's cores and PCIe, and RCCL's multi-rank latency.
    shard_proxy = None
    if world == 1 and (args.shard_proxy or collect) and len(cands) >= 16:
        import statistics

        def time_shard(idx, reps):
            align_shard(idx)                                       # warm-up of this batch size (pool blocks, scratch sizes)
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                align_shard(idx)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            return statistics.median(ts)

        every = list(range(len(cands)))
        reps = max(2, args.steps)
        t_all = time_shard(every, reps)
        loc = align_shard(mine)
        tg = []
        for _ in range(5):
            t1 = time.perf_counter()
            lc.allgather_edges_rccl(comm, loc, mine, costs)
            tg.append(time.perf_counter() - t1)
        t_gather = statistics.median(tg)
        per_world, predicted = {}, {}
        for W in (2, 4, 8):
            tr = [time_shard(list(lc.shard(costs, W, r)), reps) for r in range(W)]
            per_world[str(W)] = dict(shard_pairs=[int(len(lc.shard(costs, W, r))) for r in range(W)], shard_ms=[round(t * 1e3, 3) for t in tr],
                                     max_shard_ms=max(tr) * 1e3)
            predicted[str(W)] = t_all / (max(tr) + t_gather)
        shard_proxy = dict(pairs=len(cands), t_all_pairs_ms=t_all * 1e3, allgather_world1_ms=t_gather * 1e3, per_world=per_world,
                           predicted_speedup=predicted, repetitions=reps,
                           how="rank r's LPT shard (pgicp_shard_pairs) of a W-rank job aligned alone on this GPU, every r, W in {2, 4, 8}; "
                               "predicted_speedup(W) = T(all pairs) / (max_r T(shard) + world-1 all-gather latency); medians")
    seen = sorted(set(int(v) - 1 for v in np.asarray(edges["reserved"])[:, 1] if v > 0))
    comm_world, comm_rank = comm.info() if comm is not None else (world, rank)
    check_all_ranks_reported(seen, comm_world, world)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle on a bounded sample of the same pairs: ICP::operator() (index build + loop) + the residual chain
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle
        import statistics
        o = Oracle(np.float32)
        sample = list(range(0, len(cands), max(1, len(cands) // 8)))[:8]
        times = []
        for p_ in sample:
            c_ = cands[p_]
            rd, rx, rn = (c_.reading.cpu().numpy(), c_.ref_xyz.cpu().numpy(), c_.ref_nrm.cpu().numpy())
            t1 = time.perf_counter()
            r_ = o.icp(rd, rx, rn, c_.T_init, **CHAIN)
            if r_["status"] == 0:
                o.partial_chain(rd, rx, rn, r_["T"], **CHAIN)
            times.append(time.perf_counter() - t1)
        cpu = dict(value=1.0 / statistics.median(times), unit="pairs/s", cores=1, kind="port",
                   sample=f"{len(sample)} of the {len(cands)} pairs through the CPU oracle (k-d tree build + ICP + residual chain per pair) "
                          f"on one core, median; host has {os.cpu_count()} cores")
    out = None
    if rank == 0:
        ok = int(np.sum(edges["status"] == 0))
        out = ({
            "roofline": roofline, "cpu_baseline": cpu,
            "metric": "loop-closure candidate ICPs/sec (100k-pt keyframe vs 100k-pt candidate map)",
            "value": args.steps * len(cands) / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batched loop-closure ICP, {len(cands)} pairs of {args.n_scan}-pt clouds "
                                   f"(BASELINE.json configs[4]), index build + ICP + residual check per pair, "
                                   f"all-gather of 512-byte edge records", "pair_chunk": args.pair_chunk,
                       "parallelism": f"pairs sharded over {world} rank(s), one ncclAllGather (pgicp_allgather_edges, RCCL)"},
            "pairs_ok": ok, "pairs_accepted": int(np.sum(edges["accepted"] == 1)),
            "mean_iterations": float(np.mean(edges["iterations"])),
            "rccl_ranks_seen": len(seen), "ranks_that_reported_edges": seen, "comm_world_size": comm_world,
            "edges_transport": "pgicp_allgather_edges (the C ABI's ncclAllGather over RCCL)" if comm is not None
                               else f"torch.distributed all_gather (RCCL) -- the C ABI's communicator could not be made: {comm_error}",
            "shard_proxy": shard_proxy,
            "pairs_per_s_one_gpu_same_run": single,
            "speedup_vs_one_gpu": (args.steps * len(cands) / elapsed) / single if single else None})
    if comm is not None:
        comm.close()
    ctx.close()
    for c_ in extra_ctx:
        c_.close()
    if collect:
        return out
    if out is not None:
        emit(out)
    dist_end()


def build_drive(n_scans, n_pts, step, cache_dir="/tmp"):
    """Scans of synth.make_drive (BASELINE configs[2]) with the ray casting spread over host cores, cached."""
    from pgslam_amd import synth
    path = os.path.join(cache_dir, f"pgslam_amd_drive_{n_scans}_{n_pts}_{step}.npz")
    import math
    u = synth.uniform01(synth.DRIVE_SEED, 3 * n_scans)
    poses, odom = [], []
    T_o = None
    for s in range(n_scans):                       # same poses / odometry as synth.make_drive
        T = synth.se3(x=-40.0 + step * s, y=0.8 * math.sin(0.05 * s), yaw=math.radians(3.0) * math.sin(0.15 * s))
        if s == 0:
            T_o = T.copy()
        else:
            err = synth.se3(x=0.02 * (2 * u[3 * s] - 1), y=0.02 * (2 * u[3 * s + 1] - 1),
                            yaw=math.radians(0.15) * (2 * u[3 * s + 2] - 1))
            T_o = T_o @ (synth.se3_inv(poses[-1]) @ T) @ err
        poses.append(T)
        odom.append(T_o.copy())
    if os.path.exists(path):
        z = np.load(path)
        return poses, odom, list(z["xyz"]), list(z["nrm"])
    import multiprocessing as mp
    rings = 64 if n_pts >= 50_000 else 16
    jobs = [("drive", 7000 + s, n_pts, rings, poses[s]) for s in range(n_scans)]
    with mp.get_context("fork").Pool(max(1, min(len(jobs), os.cpu_count() or 1, 32))) as pool:
        res = pool.map(_gen_scan, jobs)
    xyz, nrm = [r[0] for r in res], [r[1] for r in res]
    try:
        np.savez(path + ".tmp.npz", xyz=np.stack(xyz), nrm=np.stack(nrm))
        os.replace(path + ".tmp.npz", path)
    except OSError:
        pass
    return poses, odom, xyz, nrm


def _gen_city_scan(args):
    from pgslam_amd import synth
    block, s, n_pts, pose = args
    global _CITY
    try:
        city = _CITY[block]
    except (NameError, KeyError):
        _CITY = {block: synth.make_city(block)}
        city = _CITY[block]
    return synth.make_city_scan(city, pose, n_pts, s)


def build_sequence(n_scans, n_pts, step, cache_dir="/tmp"):
    """BASELINE configs[3]: the KITTI-00-shaped drive of synth.city_route as a sequence file for tools/slam_run
    (format in its header).  Ray casting spread over host cores, written scan by scan, cached."""
    from pgslam_amd import synth
    path = os.path.join(cache_dir, f"pgslam_amd_seq_{n_scans}_{n_pts}_{step}.bin")
    if os.path.exists(path) and os.path.getsize(path) == 12 + n_scans * (256 + 24 * n_pts):
        return path
    import multiprocessing as mp
    block, poses = synth.city_route(n_scans, step)
    odom = synth.city_odometry(poses)
    nproc = max(1, min(os.cpu_count() or 1, 32))
    with open(path + ".tmp", "wb") as f, mp.get_context("fork").Pool(nproc) as pool:
        f.write(np.array([0x51534750, n_scans, n_pts], dtype=np.int32).tobytes())
        jobs = ((block, s, n_pts, poses[s]) for s in range(n_scans))
        for s, (xyz, nrm) in enumerate(pool.imap(_gen_city_scan, jobs, chunksize=8)):
            f.write(np.ascontiguousarray(poses[s], dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(odom[s], dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(xyz, dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(nrm, dtype=np.float32).tobytes())
    os.replace(path + ".tmp", path)
    return path


def build_slam_run():
    """g++ build of the C++ driver against include/ and the in-tree libpgicp.so (always rebuilt: seconds)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "slam_run")
    lib = os.path.join(ROOT, "pgslam_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unused-local-typedefs", "-Wno-unused-variable", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "slam_run.cpp"), "-o", exe, "-L" + lib, "-lpgicp",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def read_replay(path):
    """Records written by slam_run --record: list of dicts (kind, scan, reading, ref_xyz, ref_nrm, T_init, T_out, ...)."""
    out = []
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(16), dtype=np.int32)
        assert head[0] == 0x50524750 and head[1] == 1, "not a slam_run record file"
        for _ in range(int(head[2])):
            h = np.frombuffer(f.read(32), dtype=np.int32)
            t = np.frombuffer(f.read(256), dtype=np.float64)
            ov = float(np.frombuffer(f.read(8), dtype=np.float64)[0])
            n, m = int(h[2]), int(h[3])
            rd = np.frombuffer(f.read(12 * n), dtype=np.float32).reshape(n, 3)
            rx = np.frombuffer(f.read(12 * m), dtype=np.float32).reshape(m, 3)
            rn = np.frombuffer(f.read(12 * m), dtype=np.float32).reshape(m, 3)
            out.append(dict(kind=int(h[0]), scan=int(h[1]), iterations=int(h[4]), converged=int(h[5]), status=int(h[6]),
                            max_iter_reached=int(h[7]), overlap=ov, T_init=t[:16].reshape(4, 4), T_out=t[16:].reshape(4, 4),
                            reading=rd, ref_xyz=rx, ref_nrm=rn))
    return out


def recorded_traffic(name, **match):
    """HBM-fabric bytes per fast-matcher launch of a workload, from profiles/<name>.json (written by tools/pmc_traffic.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command; MI355X_MICROARCH.md's unit and gfx950 corrections).
    Returns the fields to merge into a roofline object: the record is used only when its workload keys equal this run's."""
    path = os.path.join(ROOT, "profiles", name + ".json")
    try:
        tj = json.load(open(path))
    except (OSError, ValueError):
        return dict(traffic=None)
    if any(tj.get(k) != v for k, v in match.items()):
        return dict(traffic=None, traffic_source="profiles/%s.json is of another workload size" % name)
    return dict(traffic=tj.get("hbm_bytes_per_launch"), traffic_uncorrected=tj.get("hbm_bytes_per_launch_uncorrected"),
                traffic_source="NOT measured in this run: profiles/%s.json, separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                               "this command (tools/pmc_round.sh); %s" % (name, tj.get("note", "")))


def slam_roofline(res, points=None):
    """Fast-matcher roofline of a slam_run pass made with PGICP_PROFILE_ALL=1: 20 N + 12 M per active problem
    over every context of the facade (pgicp_profile_process), HIP events on the contexts' streams."""
    k = (res or {}).get("knn_profile")
    if not k or not k["launches"]:
        return None
    alg = 20.0 * k["reading_points"] + 12.0 * k["map_points"]
    avg_s = k["total_ms"] * 1e-3 / k["launches"]
    ach = alg / k["launches"] / avg_s / 1e9
    return dict(bound="hbm", kernel="knn_grid", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                avg_launch_us=avg_s * 1e6, launches=k["launches"], algorithmic_bytes_per_launch=alg / k["launches"],
                active_problems_per_launch=k["problems"] / k["launches"],
                **recorded_traffic("knn_traffic_slam", n_scan=points, batch=1),
                note="one 10k-pt scan against a 30k-pt local map per launch: launch-latency-bound, the kernel cannot fill the chip; "
                     "measured in the recorded pass (every context profiling), not in the timed one")


def main_slam(args, collect=False):
    """BASELINE configs[3]: full pose-graph SLAM on the synthetic KITTI-00-shaped sequence through the C++ facade
    (host clouds in, as a pgslam user feeds them); pose-graph solve on the host.  One step = the whole sequence.
    N > 1: independent replicas (one sequence per rank, each process pinned to its GPU)."""
    import subprocess
    world, rank, local_rank = ranks()
    distributed = dist_begin("gloo", local_rank)          # only a barrier and a max over ranks: host side
    if distributed:
        import torch
        import torch.distributed as dist
    if rank == 0:
        seq = build_sequence(args.slam_scans, args.slam_points, args.slam_step)
        exe = build_slam_run()
    if distributed:
        dist.barrier()
    if rank != 0:
        seq = build_sequence(args.slam_scans, args.slam_points, args.slam_step)
        exe = os.path.join(ROOT, "tools", "slam_run")
    if args.prepare_only:
        return
    env = dict(os.environ)
    if distributed:
        env["HIP_VISIBLE_DEVICES"] = str(local_rank)
    rec = f"/tmp/pgslam_amd_replay_{rank}.bin"

    def run(record):
        # --passes P: one process, P passes over the sequence with a fresh facade each; the first is the process's warm-up (code
        # object load, first allocations, page cache), `slam_s_median_timed` is the median of the others
        cmd = [exe, seq, "--filters", args.slam_filters, "--passes", str(1 if record else args.slam_passes)] + \
              (["--record", str(args.slam_record), rec] if record else [])
        t0 = time.perf_counter()
        out = subprocess.run(cmd, env=dict(env, PGICP_PROFILE_ALL="1") if record else env, capture_output=True, text=True, check=True)
        return json.loads(out.stdout.strip().splitlines()[-1]), time.perf_counter() - t0

    for _ in range(args.warmup):
        run(False)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    res = None
    slam_s = 0.0
    for _ in range(args.steps):
        res, _ = run(False)
        slam_s += res["slam_s_median_timed"]
    if distributed:
        dist.barrier()
        t = torch.tensor([slam_s], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        slam_s = float(t.item())
    # the multi-thread flavour (pgslam::PoseGraphSlamMT: input stage, localizer, loop closer and optimiser on their own threads) on
    # the same sequence, free running -- reported next to the single-thread figure, never `value`
    mt = None
    if rank == 0 and not distributed:
        try:
            # (as for the single-thread legs: passes inside one process, the first one the warm-up, the rate the median of the rest)
            o_mt = subprocess.run([exe, seq, "--filters", args.slam_filters, "--mt", "--passes", str(max(1, args.slam_passes))], env=env, capture_output=True, text=True, check=True)
            d_mt = json.loads(o_mt.stdout.strip().splitlines()[-1])
            mt = {k: d_mt.get(k) for k in ("scans_per_s", "wall_s", "keyframes", "loops_closed", "loop_batches", "largest_loop_batch", "loop_batches_on_device",
                                           "tracking_error_last_m", "localizer_thread_s", "passes", "pass_wall_s")}
        except Exception as e:                      # reported, not hidden
            mt = dict(error=f"{type(e).__name__}: {e}")
    # one more pass that records ICP calls, replayed through the CPU oracle: parity evidence + the CPU figure
    cpu = None
    replay = None
    res_prof = None
    if rank == 0 and not args.no_cpu_baseline:
        res_prof, _ = run(True)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle
        o = Oracle(np.float32)
        recs = read_replay(rec)
        worst_t = worst_r = 0.0
        same_iters = 0
        t_cpu = 0.0
        for r in recs:
            t1 = time.perf_counter()
            ref = o.icp(r["reading"], r["ref_xyz"], r["ref_nrm"], r["T_init"], **CHAIN)
            t_cpu += time.perf_counter() - t1
            d = np.linalg.inv(ref["T"]) @ r["T_out"]
            worst_t = max(worst_t, float(np.linalg.norm(d[:3, 3])))
            # (rotation angle from the skew part: arccos(trace) turns the 6e-8 rounding of the facade's float matrices into 3e-4 rad)
            worst_r = max(worst_r, float(np.linalg.norm([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]]) / 2.0))
            same_iters += int(ref["iterations"] == r["iterations"])
        replay = dict(calls=len(recs), loop_closure_calls=sum(1 for r in recs if r["kind"] == 1), worst_translation_m=worst_t,
                      worst_rotation_rad=worst_r, same_iteration_count=same_iters)
        cpu = dict(value=len(recs) / t_cpu if t_cpu > 0 else 0.0, unit="ICP calls/s", cores=1, kind="port",
                   sample=f"the {len(recs)} ICP calls recorded from the run (scan-to-local-map and loop closure, index build included), "
                          f"replayed one after the other through the CPU oracle on one core; host has {os.cpu_count()} cores")
    out = None
    if rank == 0:
        n = res["scans"] - 1
        out = ({
            "metric": "scans/sec through full pose-graph SLAM (synthetic KITTI-00-shaped sequence, loop closures, host pose-graph solve)",
            "value": args.steps * n * world / slam_s, "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": slam_s * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"full pose-graph SLAM, {res['scans']} scans of {res['points_per_scan']} pts, {args.slam_step} m apart "
                                   f"(BASELINE.json configs[3]), host clouds through pgslam::PoseGraphSlam<float> (C++ facade), input filters: "
                                   f"{res.get('input_filters')}",
                       "parallelism": f"{world} independent replica(s), one process per GPU"},
            "slam": res, "slam_mt": mt, "replay_vs_oracle": replay, "cpu_baseline": cpu, "roofline": slam_roofline(res_prof, args.slam_points)})
    if collect:
        return out
    if out is not None:
        emit(out)
    dist_end()


def main_stream(args, collect=False):
    """BASELINE configs[2]: a scan feed through the streaming local mapper -- ICP against a sliding,
    device-resident map of `--capacity` keyframes (20 x 100k = 2M points), new keyframe when the overlap
    drops below 0.8, the next map assembled and indexed on a background context while scans keep
    aligning.  A "step" is one pass of every stream over the timed scans; `--streams` independent
    vehicles (one mapper, one host thread, one HIP stream each) share the GPU.  Scans are staged in
    HBM before the timed region (the PCIe-inclusive figure is reported separately)."""
    import torch
    import torch.distributed as dist
    from pgslam_amd import icp
    from pgslam_amd.local_mapper import Keyframe, LocalMapperConfig, StreamingLocalMapper

    world, rank, local_rank = ranks()
    if not args.prepare_only:
        require_device(local_rank)
    distributed = dist_begin("nccl", local_rank) if not args.prepare_only else False
    dev = torch.device("cuda", local_rank)
    n_prime = args.capacity - 1
    n_total = n_prime * args.prime_stride + args.stream_scans
    if rank == 0:
        drive = build_drive(n_total, args.n_scan, args.stream_step)
    if distributed:
        dist.barrier()
    if rank != 0:
        drive = build_drive(n_total, args.n_scan, args.stream_step)
    poses, odom, xyz, nrm = drive
    if args.prepare_only:
        return
    first = n_prime * args.prime_stride
    # the map so far is taken as accurate: keyframes sit at their true poses and the odometry frame is
    # re-based on the truth at the first timed scan, so the final error below is the mapper's own
    rebase = poses[first] @ np.linalg.inv(odom[first])
    odom = [poses[s] if s < first else rebase @ odom[s] for s in range(n_total)]
    d_xyz = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in xyz]
    d_nrm = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in nrm]

    class Vehicle:
        def __init__(self):
            self.ctx = icp.Context(local_rank, **CHAIN)
            self.builder = icp.Context(local_rank, **CHAIN) if not args.sync_rebuild else None
            self.new()

        def new(self):
            cfg = LocalMapperConfig(capacity=args.capacity, overlap_threshold=0.8, chain=dict(CHAIN),
                                    async_rebuild=not args.sync_rebuild)
            self.m = StreamingLocalMapper(self.ctx, cfg, builder=self.builder)
            # the window the vehicle would have after the drive so far: every prime_stride-th earlier scan
            for k in range(n_prime):
                s = k * args.prime_stride
                self.m.window.append(Keyframe(self.m.next_kf_id, d_xyz[s], d_nrm[s], odom[s].copy()))
                self.m.next_kf_id += 1
            self.m.process(odom[first], d_xyz[first], d_nrm[first])        # becomes the reference keyframe; builds the map

        def run(self, out):
            its, conv = 0, 0
            for s in range(first + 1, n_total):
                self.m.process(odom[s], d_xyz[s], d_nrm[s])
                st = self.m.last_stats
                its += st["iterations"]
                conv += int(st["status"] == 0 and st["converged"])
            out.append((its, conv, len(self.m.keyframe_scans) - 1, self.m.rebuilds,
                        float(np.linalg.norm((np.linalg.inv(poses[n_total - 1]) @ self.m.T_world_robot)[:3, 3]))))

        def reset(self):
            self.m.close()
            self.new()

    per_step = (n_total - first - 1) * args.streams

    class Fleet:
        """--fleet: all vehicles stepped together, one device batch per time step (StreamingFleet)."""
        def __init__(self):
            from pgslam_amd.local_mapper import StreamingFleet
            self.ctx = icp.Context(local_rank, **CHAIN)
            self.builder = None
            self.StreamingFleet = StreamingFleet
            self.new()

        def new(self):
            cfg = LocalMapperConfig(capacity=args.capacity, overlap_threshold=0.8, chain=dict(CHAIN))
            self.f = self.StreamingFleet(self.ctx, args.streams, cfg)
            for m in self.f.mappers:
                for k in range(n_prime):
                    s = k * args.prime_stride
                    m.window.append(Keyframe(m.next_kf_id, d_xyz[s], d_nrm[s], odom[s].copy()))
                    m.next_kf_id += 1
            self.f.step([odom[first]] * args.streams, [d_xyz[first]] * args.streams, [d_nrm[first]] * args.streams)
            self.m = self.f.mappers[0]

        def run(self, out):
            its, conv = 0, 0
            for s in range(first + 1, n_total):
                self.f.step([odom[s]] * args.streams, [d_xyz[s]] * args.streams, [d_nrm[s]] * args.streams)
                for m in self.f.mappers:
                    its += m.last_stats["iterations"]
                    conv += int(m.last_stats["status"] == 0 and m.last_stats["converged"])
            m = self.f.mappers[0]
            out.append((its, conv, len(m.keyframe_scans) - 1, m.rebuilds,
                        float(np.linalg.norm((np.linalg.inv(poses[n_total - 1]) @ m.T_world_robot)[:3, 3]))))

        def reset(self):
            self.f.close()
            self.new()

    vehicles = [Fleet()] if args.fleet else [Vehicle() for _ in range(args.streams)]
    vehicles[0].ctx.debug_counters()                # (reading them resets them: the guess misses below are this leg's)

    def step():
        outs = [[] for _ in vehicles]
        ths = [threading.Thread(target=v.run, args=(o,)) for v, o in zip(vehicles, outs)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return [o[0] for o in outs]

    for _ in range(args.warmup):
        step()
        for v in vehicles:
            v.reset()
    import gc
    gc.collect()
    gc.freeze()                     # (see main())
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = 0.0
    pass_s = []
    for k in range(args.steps):
        t0 = time.perf_counter()
        res = step()
        torch.cuda.synchronize()
        pass_s.append(time.perf_counter() - t0)
        elapsed += pass_s[-1]
        if k + 1 < args.steps:
            for v in vehicles:                      # rewinding the vehicles is not part of the feed
                v.reset()
    if distributed:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- roofline of the dominant kernel (the fast matcher), SURVEY.md 8(d): per problem and launch 20 N + 12 M with M the
    #      sliding map's points; HIP events on the serving context's stream over a replay of one vehicle's (or the fleet's) drive
    roofline = None
    if not args.no_profile:
        v = vehicles[0]
        v.reset()
        v.ctx.profile_reset()
        v.ctx.profile_enable(True)
        v.run([])
        v.ctx.profile_enable(False)
        k = v.ctx.profile()["knn_grid"]
        if k["launches"]:
            m_map = args.capacity * args.n_scan
            alg = 20.0 * k["units"] + 12.0 * m_map * k["problems"]
            avg_s = k["total_ms"] * 1e-3 / k["launches"]
            achieved = alg / k["launches"] / avg_s / 1e9
            roofline = dict(bound="hbm", kernel="knn_grid", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                            avg_launch_us=avg_s * 1e6, launches=k["launches"],
                            algorithmic_bytes_per_launch=alg / k["launches"], active_problems_per_launch=k["problems"] / k["launches"],
                            **recorded_traffic("knn_traffic_stream", n_scan=args.n_scan, n_map=m_map, batch=args.streams if args.fleet else 1),
                            note="one small launch per vehicle (or one per fleet step): the kernel cannot fill the chip; see DESIGN.md section 5")
    # ---- PCIe-inclusive companion (never `value`): the caller owns HOST scans (Localizer.hpp:103-126); the mapper uploads
    #      scan k + 1 on the context's copy stream while scan k aligns (LocalizerMT.hpp:27-40 has it queued by then)
    host_input = None
    if not args.no_host_input and not args.fleet and args.streams == 1:
        v = vehicles[0]
        block = v.ctx.host_alloc((n_total - first,) + xyz[0].shape, np.float32)     # a sensor driver's pinned ring buffer
        for s_ in range(first, n_total):
            block[s_ - first] = xyz[s_]
        rates = {}
        for label, src, pinned in (("pinned", [block[s_ - first] for s_ in range(first, n_total)], True),
                                   ("pageable", [np.ascontiguousarray(xyz[s_]) for s_ in range(first, n_total)], False)):
            best = 0.0
            for _ in range(1 + max(1, args.steps)):                                 # the first pass is the pipeline's warm-up
                v.reset()
                v.m.pinned_sources = pinned
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                v.m.stage(src[1])
                for s_ in range(first + 1, n_total):
                    nxt = src[s_ + 1 - first] if s_ + 1 < n_total else None
                    v.m.process(odom[s_], src[s_ - first], nrm[s_], next_xyz=nxt)
                torch.cuda.synchronize()
                best = max(best, (n_total - first - 1) / (time.perf_counter() - t0))
            rates[label] = best
        v.ctx.host_free(block)
        dev_rate = args.steps * per_step / elapsed
        host_input = dict(pinned_scans_per_s=rates["pinned"], pageable_scans_per_s=rates["pageable"],
                          pinned_over_device_resident=rates["pinned"] / dev_rate, pageable_over_device_resident=rates["pageable"] / dev_rate,
                          bytes_per_scan=int(args.n_scan * 12),
                          how="host scans through StreamingLocalMapper.process(..., next_xyz=): scan k + 1 uploaded on the copy stream "
                              "(pgicp_upload_f32) while scan k aligns; best of the passes after the pipeline's first")
    # ---- CPU baseline: the oracle on the same feed, one thread per vehicle (a vehicle's scans are sequential), bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle
        o = Oracle(np.float32)
        kf = [first] + [kk * args.prime_stride for kk in range(n_prime)]          # reference keyframe first
        inv_ref = np.linalg.inv(poses[first])
        t0 = time.perf_counter()
        mx, mn = o.build_local_map([xyz[s_] for s_ in kf], [nrm[s_] for s_ in kf], [inv_ref @ odom[s_] for s_ in kf])
        m_o = o.map_create(mx, mn, center=True, use_kdtree=True)
        t_build = time.perf_counter() - t0
        n_thr = max(1, min(args.streams, os.cpu_count() or 1))
        sample = list(range(first + 1, min(n_total, first + 1 + 6)))
        done = []

        def work():
            for s_ in sample:
                done.append(o.icp_map(m_o, xyz[s_], inv_ref @ odom[s_], **CHAIN))

        th = [threading.Thread(target=work) for _ in range(n_thr)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        o.map_free(m_o)
        cpu = dict(value=sum(1 for r in done if r["status"] == 0 and r["converged"]) / dt, unit="scans/s", cores=n_thr, kind="port",
                   sample=f"the first {len(sample)} timed scans per vehicle against the first sliding map ({len(kf)} keyframes, {mx.shape[0]} points), "
                          f"kd-tree oracle, one thread per vehicle ({n_thr}); the map's index build ({t_build:.1f} s, once per keyframe on the CPU) excluded",
                   host_cores=os.cpu_count(), index_build_s=t_build, mean_iterations=float(np.mean([r["iterations"] for r in done])))
    out = None
    if rank == 0:
        its = sum(r[0] for r in res)
        conv = sum(r[1] for r in res)
        mode = ("one device batch per time step (fleet; the vehicles are replicas of ONE drive: same scans, odometry and keyframes)" if args.fleet else
                ("synchronous" if args.sync_rebuild else "background") + " map rebuild, one host thread per vehicle")
        out = ({
            "metric": "streamed scans/sec through the local mapper (100k-pt scans, sliding 2M-pt device-resident map)",
            "value": args.steps * per_step * world / elapsed, "unit": "scans/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"streaming local mapper, {args.stream_scans - 1} timed scans of {args.n_scan} pts per vehicle, "
                                   f"{args.capacity}-keyframe sliding map ({args.capacity * args.n_scan} pts, BASELINE.json configs[2]), "
                                   f"{args.streams} vehicle(s) per GPU, {mode}",
                       "parallelism": f"{world} GPU(s) x {args.streams} independent vehicles (replicas)"},
            "ms_per_scan_per_vehicle": elapsed * 1e3 / (args.steps * (n_total - first - 1)),
            # `value` is the contract's figure (all timed passes / their time); the passes one by one, and their median
            "selection_guess_misses_per_scan": vehicles[0].ctx.debug_counters()[3] / max(1.0, float(per_step) / max(1, args.streams) * (args.steps + args.warmup)),
            "scans_per_s_each_pass": [per_step * world / t for t in pass_s],
            "scans_per_s_median_pass": float(np.median([per_step * world / t for t in pass_s])),
            "mean_iterations": its / per_step, "converged_fraction": conv / per_step,
            "new_keyframes_per_vehicle": res[0][2], "map_rebuilds_per_vehicle": res[0][3], "final_position_error_m": res[0][4],
            "host_input": host_input, "roofline": roofline, "cpu_baseline": cpu})
    for v in vehicles:
        v.m.close()
        v.ctx.close()
        if v.builder:
            v.builder.close()
    if collect:
        return out
    if out is not None:
        emit(out)
    dist_end()


def main_f64(args, collect=False):
    """PointMatcher<double> (the reference instantiates it on equal footing, /root/reference/tests/instantiation.cpp:12-18):
    the headline's workload -- the same scans, map, initial guesses and chain -- with every scalar a double.  Algorithmic
    bytes double with the scalars: the matcher kernel 40 N + 24 M per active problem."""
    import torch
    from pgslam_amd import icp, synth
    world, rank, local_rank = ranks()
    require_device(local_rank)
    dev = torch.device("cuda", local_rank)
    w = build_workload(args.n_scan, args.n_map, args.queries)
    chain = dict(CHAIN)
    ctx = icp.Context(local_rank, **chain, matcher=icp.MATCHER_GRID, check_every=args.check_every)
    d_map_xyz = torch.from_numpy(w.map_xyz.astype(np.float64)).to(dev)
    d_map_nrm = torch.from_numpy(w.map_nrm.astype(np.float64)).to(dev)
    d_scans = [torch.from_numpy(s.astype(np.float64)).to(dev) for s in w.scans_xyz]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    map_id = ctx.set_map(d_map_xyz, d_map_nrm, center=True)
    t_setmap = time.perf_counter() - t0
    B = args.batch
    readings = [d_scans[b % len(d_scans)] for b in range(B)]
    T_inits = [w.T_truth[b % len(d_scans)] @ synth.perturbation(1000 * rank + b) for b in range(B)]

    def step():
        return ctx.align_batch(map_id, readings, T_inits, raise_on_error=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    converged, iters, last = 0, [], None
    for _ in range(args.steps):
        T, st = step()
        converged += sum(1 for s in st if s["status"] == 0 and s["converged"])
        iters += [s["iterations"] for s in st]
        last = (T, st)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    T, st = last
    err_t = [float(np.linalg.norm((np.linalg.inv(w.T_truth[b % len(d_scans)]) @ T[b])[:3, 3])) for b in range(B) if st[b]["status"] == 0]
    # roofline of the matcher kernel: HIP events on the context's stream around every launch of a replay of the timed steps
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(args.steps):
        step()
    ctx.profile_enable(False)
    k = ctx.profile()["knn_grid"]
    roofline = None
    if k["launches"]:
        alg = 40.0 * k["units"] + 24.0 * args.n_map * k["problems"]
        avg_s = k["total_ms"] * 1e-3 / k["launches"]
        roofline = dict(bound="hbm", kernel="knn_grid<double>", achieved=alg / k["launches"] / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=alg / k["launches"] / avg_s / 1e9 / HBM_PEAK_GBS, avg_launch_us=avg_s * 1e6, launches=k["launches"],
                        active_problems_per_launch=k["problems"] / k["launches"], algorithmic_bytes_per_launch=alg / k["launches"],
                        **recorded_traffic("knn_traffic_f64", n_scan=args.n_scan, n_map=args.n_map, batch=B))
    cpu = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle
        o = Oracle(np.float64)
        t0 = time.perf_counter()
        m = o.map_create(w.map_xyz.astype(np.float64), w.map_nrm.astype(np.float64), center=True, use_kdtree=True)
        t_build = time.perf_counter() - t0
        n_cpu = 4
        t0 = time.perf_counter()
        rs = [o.icp_map(m, w.scans_xyz[q].astype(np.float64), w.T_init[q], **CHAIN) for q in range(n_cpu)]
        dt = time.perf_counter() - t0
        o.map_free(m)
        cpu = dict(value=sum(1 for r in rs if r["status"] == 0 and r["converged"]) / dt, unit="scans/s", cores=1, kind="port",
                   sample=f"{n_cpu} of the benchmark's scans vs the 1M-pt map through the float64 CPU oracle (k-d tree port of the chain) on one "
                          f"core; index build ({t_build:.2f} s) excluded; host has {os.cpu_count()} cores", index_build_s=t_build)
    ctx.destroy_map(map_id)
    ctx.close()
    del d_map_xyz, d_map_nrm, d_scans, readings
    torch.cuda.empty_cache()
    out = dict(metric="ICP-converged scans/sec at 100k-pt scan vs 1M-pt local map, PointMatcher<double>", value=converged / elapsed, unit="scans/s",
               n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=elapsed * 1e3 / args.steps, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload=f"scan-to-map ICP, {args.n_scan}-pt scan vs {args.n_map}-pt map, <=30 iterations (BASELINE.json configs[1]) "
                                    f"with double scalars", batch_scans_per_step=B, parallelism="1 replica"),
               mean_iterations=float(np.mean(iters)), converged_fraction=converged / max(1, args.steps * B), set_map_ms=t_setmap * 1e3,
               median_translation_error_m=float(np.median(err_t)) if err_t else None, roofline=roofline, cpu_baseline=cpu)
    if collect:
        return out
    emit(out)


def compact_leg(d, wall_s):
    """What a leg contributes to the headline's line: the figure, its roofline and its CPU baseline, a few facts."""
    if d is None:
        return None
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config", "roofline", "cpu_baseline", "dtype",
            "set_map_ms", "median_translation_error_m",
            "mean_iterations", "converged_fraction", "final_position_error_m", "host_input", "new_keyframes_per_vehicle", "map_rebuilds_per_vehicle",
            "pairs_ok", "pairs_accepted", "rccl_ranks_seen", "ranks_that_reported_edges", "comm_world_size", "edges_transport",
            "pairs_per_s_one_gpu_same_run", "speedup_vs_one_gpu", "shard_proxy", "slam_mt", "replay_vs_oracle", "scans_per_s_each_pass", "scans_per_s_median_pass", "selection_guess_misses_per_scan")
    out = {k: d[k] for k in keep if k in d}
    if "slam" in d:
        out["slam"] = {k: d["slam"].get(k) for k in ("scans", "points_per_scan", "keyframes", "loops_closed", "loop_candidates_tried",
                                                     "map_rebuilds", "mean_icp_iterations", "tracking_error_rms_m", "localizer_host_s",
                                                     "input_filters", "device_input_stages", "device_readings_used", "device_map_rebuilds", "points_after_filters_last_scan",
                                                     "passes", "pass_slam_s", "slam_s_median_timed", "icp_call_s", "loop_candidates_assembled_on_device", "keyframes_resident")}
    r = out.get("roofline")
    if r:
        out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches",
                                             "active_problems_per_launch", "algorithmic_bytes_per_launch") if k in r}
    out["leg_wall_s"] = wall_s
    return out


def workload_legs(args, world, rank):
    """BASELINE.json configs[2], [3], [4] next to the headline, bounded: the streaming local mapper (one vehicle, 2M-pt
    sliding map), full SLAM (the 4 500-scan sequence, recorded ICP calls replayed through the oracle) -- both replicas,
    run at N = 1 only -- and the batched loop-closure ICP (512 pairs, at every N: sharded over the ranks, one
    pgicp_allgather_edges over RCCL; at N > 1 also aligned by rank 0 alone, for the strong-scaling ratio)."""
    if args.no_workloads or args.fixed_iters or args.matcher != "grid":
        return None, None
    import copy
    legs = {}

    def run(name, fn, **over):
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            d = fn(a, collect=True)
        except Exception as e:                      # a leg must not take the headline down with it
            if world > 1:
                raise                               # (but the ranks of a job must not part ways)
            legs[name] = dict(error=f"{type(e).__name__}: {e}")
            return None
        c = compact_leg(d, time.perf_counter() - t0)
        if c is not None:
            legs[name] = c
        return c

    lc_leg = run("loop_closure", main_loopclosure, steps=2, warmup=1, pairs=512, pair_chunk=512)
    if world == 1:
        run("f64", main_f64, steps=5, warmup=2)
        run("stream", main_stream, steps=3, warmup=1, streams=1, fleet=False)
        run("slam", main_slam, steps=1, warmup=0)
        # the same facade at SENSOR size: 100 k-pt scans (64 rings, an HDL-64E's), a shorter drive (the sequence file is 2.4 MB a
        # scan), the input filters a driver would configure -- every scan through the device input stage
        run("slam_100k", main_slam, steps=1, warmup=0, slam_scans=600, slam_points=100_000, slam_filters="sensor", slam_record=8)
    return (legs if rank == 0 else None), lc_leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="independent scans aligned per step")
    ap.add_argument("--queries", type=int, default=64, help="distinct query scans generated (cycled to fill the batch)")
    ap.add_argument("--n-scan", type=int, default=100_000)
    ap.add_argument("--n-map", type=int, default=1_000_000)
    ap.add_argument("--fixed-iters", action="store_true", help="disable the Differential checker: exactly 30 iterations")
    ap.add_argument("--no-fixed30", action="store_true",
                    help="skip the companion figure: by default, after the timed run, one step is also timed with the Differential "
                         "checker disabled (exactly 30 iterations per scan, SURVEY.md section 8(d)) and reported as `fixed_30_iterations`; "
                         "pass this for a kernel trace that should hold only the metric's launches")
    ap.add_argument("--with-fixed30", action="store_true", help="(the default now; kept so that older command lines still parse)")
    ap.add_argument("--matcher", choices=["grid", "brute"], default="grid")
    ap.add_argument("--sum-order", choices=["sorted", "scan"], default="sorted",
                    help="pgicp_params.sum_order: the order the pairs enter the reduction tree in (sorted: the library's sorting order of "
                         "the reading, all loads coalesced -- the default; scan: the caller's reading order, results a function of the inputs alone)")
    ap.add_argument("--grid-cell", type=float, default=0.0)
    ap.add_argument("--check-every", type=int, default=1,
                    help="the host looks at the device-side convergence flag every this many iterations (results do not depend on it; "
                         "2 is +0.7 %% but adds an empty launch per step, which the per-launch roofline accounting would have to explain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64, help="scans of the CPU baseline's batch (at least one per host core)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-host-input", action="store_true",
                    help="skip the PCIe-inclusive companion figure: by default the timed steps are repeated with the scans in HOST "
                         "memory (pinned, and pageable), uploaded one step ahead on the context's copy stream (pgicp_upload_f32) while "
                         "the current step aligns; reported as `host_input`, never as `value`")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts (HIP streams, one host thread each) the batch is split over; each keeps its own "
                         "resident copy of the map.  >1 overlaps one sub-batch's convergence tail with another's head")
    ap.add_argument("--capacity", type=int, default=20, help="stream: keyframes in the sliding map")
    ap.add_argument("--stream-scans", type=int, default=41, help="stream: scans per vehicle (the first becomes a keyframe, untimed)")
    ap.add_argument("--stream-step", type=float, default=0.35, help="stream: metres travelled between scans (10 Hz at 3.5 m/s)")
    ap.add_argument("--prime-stride", type=int, default=3, help="stream: earlier scans between the keyframes that pre-fill the window")
    ap.add_argument("--sync-rebuild", action="store_true", help="stream: rebuild the map in line, as the reference does")
    ap.add_argument("--fleet", action="store_true", help="stream: step all vehicles together, one device batch per time step")
    ap.add_argument("--slam-scans", type=int, default=4500, help="slam: scans of the sequence (KITTI-00 has 4541)")
    ap.add_argument("--slam-points", type=int, default=10_000, help="slam: points per scan (16 rings)")
    ap.add_argument("--slam-step", type=float, default=0.8, help="slam: metres between scans (10 Hz at 8 m/s)")
    ap.add_argument("--slam-filters", choices=["identity", "sensor"], default="identity",
                    help="slam: the localizer's input filters (sensor: RemoveNaN, range cut, vehicle box -- through the device input stage)")
    ap.add_argument("--slam-passes", type=int, default=4,
                    help="slam: passes over the sequence inside ONE slam_run process: the first warms the process up, `value` is the "
                         "median of the others (1: a single cold pass, as before round 5)")
    ap.add_argument("--slam-record", type=int, default=32, help="slam: ICP calls recorded for the replay through the CPU oracle")
    ap.add_argument("--workload", choices=["scan2map", "loopclosure", "stream", "slam", "f64"], default="scan2map",
                    help="scan2map = BASELINE configs[1] (the headline metric); loopclosure = configs[4]: --pairs candidate "
                         "scan pairs (100k vs 100k) sharded over the ranks, all-gather of the SE(3) edges")
    ap.add_argument("--pairs", type=int, default=512)
    ap.add_argument("--pair-chunk", type=int, default=512,
                    help="pairs aligned per device batch (measured at 512 pairs on one GPU: 64 -> 4 260, 128 -> 4 820, 256 -> 5 180, 512 -> 5 440 pairs/s)")
    ap.add_argument("--lc-contexts", type=int, default=1, help="loopclosure: sub-batches of a rank's shard aligned concurrently on this many contexts")
    ap.add_argument("--shard-proxy", action="store_true",
                    help="loopclosure at N = 1: also time every rank's LPT shard of a 2-, 4- and 8-rank job alone on this GPU and report "
                         "the predicted multi-GPU speed-up (`shard_proxy`; always on in the default line's loop_closure leg)")
    ap.add_argument("--no-workloads", action="store_true",
                    help="scan2map: skip the `workloads` legs the default line carries next to the headline (stream, loop closure, "
                         "SLAM -- BASELINE.json configs[2..4], each with its own roofline and cpu_baseline, never `value`)")
    ap.add_argument("--prepare-only", action="store_true",
                    help="generate + cache the synthetic workload and exit (run this before a rocprofv3 --pmc pass: the "
                         "generator forks worker processes, which must not happen under the counter profiler)")
    args = ap.parse_args()

    # `--gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run sets WORLD_SIZE; then we ARE a rank)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.prepare_only:
        return launch_ranks(args.gpus)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world size is what runs "
              f"(n_gpus in the line)", file=sys.stderr)

    if args.workload == "stream":
        return main_stream(args)
    if args.workload == "slam":
        return main_slam(args)
    if args.prepare_only:
        build_workload(args.n_scan, args.n_map, args.queries)
        return
    if args.workload == "loopclosure":
        return main_loopclosure(args)
    if args.workload == "f64":
        return main_f64(args)

    import torch
    import torch.distributed as dist
    from pgslam_amd import icp

    t_cmd0 = time.perf_counter()
    world, rank, local_rank = ranks()
    require_device(local_rank)
    distributed = dist_begin("nccl", local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- inputs (rank 0 generates + caches, the others load the cache) ----
    if rank == 0:
        w = build_workload(args.n_scan, args.n_map, args.queries)
    if distributed:
        dist.barrier()
    if rank != 0:
        w = build_workload(args.n_scan, args.n_map, args.queries)

    chain = dict(CHAIN)
    if args.sum_order == "scan":
        chain["sum_order"] = 1
    if args.fixed_iters:
        chain.update(min_diff_rot=0.0, min_diff_trans=0.0)
    S = max(1, args.streams)
    ctxs = [icp.Context(local_rank, **chain, matcher=icp.MATCHER_GRID if args.matcher == "grid" else icp.MATCHER_BRUTE,
                        grid_cell=args.grid_cell, check_every=args.check_every) for _ in range(S)]
    ctx = ctxs[0]
    d_map_xyz = torch.from_numpy(w.map_xyz).to(dev)
    d_map_nrm = torch.from_numpy(w.map_nrm).to(dev)
    d_scans = [torch.from_numpy(s).to(dev) for s in w.scans_xyz]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    map_id = ctx.set_map(d_map_xyz, d_map_nrm, center=True)
    t_setmap = time.perf_counter() - t0
    map_ids = [map_id] + [c.set_map(d_map_xyz, d_map_nrm, center=True) for c in ctxs[1:]]

    B = args.batch
    readings = [d_scans[b % len(d_scans)] for b in range(B)]
    # distinct problems even when scans are cycled: each slot has its own initial guess
    from pgslam_amd import synth
    T_inits = [w.T_truth[b % len(d_scans)] @ synth.perturbation(1000 * rank + b) for b in range(B)]

    def step():
        if S == 1:
            return ctx.align_batch(map_id, readings, T_inits, raise_on_error=False)
        # sub-batch k -> context k (own stream, own host thread; ctypes releases the GIL)
        out = [None] * S
        def work(k):
            out[k] = ctxs[k].align_batch(map_ids[k], readings[k::S], T_inits[k::S], raise_on_error=False)
        th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        T = np.zeros((B, 4, 4))
        st = [None] * B
        for k in range(S):
            T[k::S] = out[k][0]
            st[k::S] = out[k][1]
        return T, st

    for _ in range(args.warmup):
        step()
    # the harness itself: every object alive now (torch, numpy, the workload) moves to the collector's permanent generation,
    # so that a collection triggered by a step's few thousand short-lived result objects does not walk them (at 256-320
    # problems per step a full collection fell into every step: 10-13 ms of Python per 26 ms of ICP)
    import gc
    gc.collect()
    gc.freeze()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    converged = 0
    iters = []
    last = None
    for _ in range(args.steps):
        T, st = step()
        # --fixed-iters disables the Differential checker (workload-stable figure):
        # every scan that ran its 30 iterations with status OK counts
        converged += sum(1 for s in st if s["status"] == 0 and (s["converged"] or args.fixed_iters))
        iters += [s["iterations"] for s in st]
        last = (T, st)
    fence()
    elapsed = time.perf_counter() - t0
    sel_fallbacks = ctx.debug_counters()[3]          # (since the context's creation: warm-up included)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_max = float(t.item())
        cnt = torch.tensor([converged, args.steps * B, int(np.sum(iters))], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        converged_all, scans_all, iters_all = (int(x) for x in cnt.tolist())
    else:
        elapsed_max, converged_all, scans_all, iters_all = elapsed, converged, args.steps * B, int(np.sum(iters))

    # accuracy of what was timed (not part of the metric, guards against a fast wrong answer)
    T, st = last
    err_t = [float(np.linalg.norm((np.linalg.inv(w.T_truth[b % len(d_scans)]) @ T[b])[:3, 3])) for b in range(B)
             if st[b]["status"] == 0]

    # ---- roofline: kNN kernel timed with HIP events on the context's stream, over a replay of the timed
    #      region (the same `--steps` steps once more with an event pair around every launch) ----
    roofline = None
    kern = {}
    if not args.no_profile:
        ctx.set_params(check_every=1)
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(args.steps):
            step()
        ctx.profile_enable(False)
        prof = ctx.profile()
        kname = "knn_grid" if args.matcher == "grid" else "knn_brute"
        k = prof[kname]
        if k["launches"]:
            # algorithmic bytes: 20 N + 12 M per ACTIVE problem of a launch
            alg_bytes = 20.0 * k["units"] + 12.0 * args.n_map * k["problems"]      # sum over launches
            # physically compulsory bytes: the problems of a launch share ONE resident map, read once
            shared_bytes = 20.0 * k["units"] + 12.0 * args.n_map * k["launches"]
            avg_s = k["total_ms"] * 1e-3 / k["launches"]
            achieved = alg_bytes / k["launches"] / avg_s / 1e9
            achieved_shared = shared_bytes / k["launches"] / avg_s / 1e9
            traffic = traffic_raw = traffic_src = None
            tpath = os.path.join(ROOT, "profiles", "knn_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("n_scan") == args.n_scan and tj.get("n_map") == args.n_map and tj.get("batch") == B:
                        traffic = tj.get("hbm_bytes_per_launch")
                        traffic_raw = tj.get("hbm_bytes_per_launch_uncorrected")
                        traffic_src = ("NOT measured in this run: profiles/knn_traffic.json, separate rocprofv3 --pmc FETCH_SIZE / "
                                       "WRITE_SIZE passes of this command (tools/measure_round.sh); " + str(tj.get("note", "")))
                except (OSError, ValueError):
                    pass
            # what the kernel is actually bound by (unit counters, profiles/knn_pmc.json: separate rocprofv3 --pmc passes of this
            # command) and the same fraction for the launches that had no correspondences to start from, and for the rest
            pmc = None
            ppath = os.path.join(ROOT, "profiles", "knn_pmc.json")
            if os.path.exists(ppath):
                try:
                    pmc = json.load(open(ppath))
                except (OSError, ValueError):
                    pmc = None
            split = {}
            ku = prof.get("knn_grid_unseeded")
            if kname == "knn_grid" and ku and ku["launches"] and k["launches"] > ku["launches"]:
                def frac_of(units, problems, launches, ms):
                    return (20.0 * units + 12.0 * args.n_map * problems) / launches / (ms * 1e-3 / launches) / 1e9 / HBM_PEAK_GBS
                split = dict(frac_unseeded_launches=frac_of(ku["units"], ku["problems"], ku["launches"], ku["total_ms"]),
                             avg_unseeded_launch_us=ku["total_ms"] * 1e3 / ku["launches"],
                             frac_seeded_launches=frac_of(k["units"] - ku["units"], k["problems"] - ku["problems"], k["launches"] - ku["launches"],
                                                          k["total_ms"] - ku["total_ms"]),
                             avg_seeded_launch_us=(k["total_ms"] - ku["total_ms"]) * 1e3 / (k["launches"] - ku["launches"]))
            roofline = dict(bound="hbm", kernel=kname, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                            bound_means="`bound` names the roofline the fraction is QUOTED against (BASELINE.json north_star: achieved fraction of the "
                                        "HBM roofline, SURVEY.md 8(d) algorithmic bytes); `bound_measured` is what the unit counters say limits the kernel",
                            frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_uncorrected=traffic_raw,
                            traffic_source=traffic_src,
                            bound_measured=(pmc or {}).get("bound_measured", "valu"),
                            traffic_measured_in_this_run=False,
                            valu_lane_ops_per_active_query_iteration=(pmc or {}).get("valu_lane_ops_per_active_query_iteration"),
                            bound_measured_evidence=(pmc or {"note": "profiles/knn_pmc.json not found: see DESIGN.md section 4 (VALU 68-94 % busy)"}),
                            **split,
                            achieved_shared_map=achieved_shared, frac_shared_map=achieved_shared / HBM_PEAK_GBS,
                            compulsory_bytes_per_launch_shared_map=shared_bytes / k["launches"],
                            avg_launch_us=avg_s * 1e6, launches=k["launches"], profiled_steps=args.steps,
                            algorithmic_bytes_per_launch=alg_bytes / k["launches"],
                            active_problems_per_launch=k["problems"] / k["launches"])
        for name, v in prof.items():
            if v["launches"]:
                kern[name] = dict(launches=v["launches"], total_ms=round(v["total_ms"], 4),
                                  avg_us=round(v["total_ms"] * 1e3 / v["launches"], 2))

    # ---- workload-stable companion figure: the same step with the Differential
    #      checker disabled, i.e. exactly 30 iterations per scan; reported next to the metric, never as `value`
    fixed30 = None
    if not args.no_fixed30 and not args.fixed_iters:
        for c in ctxs:
            c.set_params(min_diff_rot=0.0, min_diff_trans=0.0, check_every=args.check_every)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, st30 = step()
        torch.cuda.synchronize()
        dt30 = time.perf_counter() - t0
        fixed30 = dict(scans_per_s=sum(1 for s in st30 if s["status"] == 0) / dt30, ms_per_step=dt30 * 1e3,
                       iterations=float(np.mean([s["iterations"] for s in st30])))
        for c in ctxs:
            c.set_params(min_diff_rot=chain["min_diff_rot"], min_diff_trans=chain["min_diff_trans"])

    # ---- companion figure: the same steps with the batch split over TWO contexts (two HIP streams, two host threads, a resident
    #      copy of the map each): the last, nearly empty iterations of one half overlap the other half's work.  Never `value`:
    #      the roofline above describes the launches of ONE context, which is what the metric is quoted on.
    two_ctx = None
    if S == 1 and B >= 64 and not args.no_fixed30 and not args.fixed_iters and world == 1:
        c2 = [ctx, icp.Context(local_rank, **chain, matcher=icp.MATCHER_GRID if args.matcher == "grid" else icp.MATCHER_BRUTE,
                               grid_cell=args.grid_cell, check_every=args.check_every)]
        m2 = [map_id, c2[1].set_map(d_map_xyz, d_map_nrm, center=True)]

        def step2():
            out = [None, None]
            def work(k):
                out[k] = c2[k].align_batch(m2[k], readings[k::2], T_inits[k::2], raise_on_error=False)
            th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            return out
        step2(); step2()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n2 = 0
        for _ in range(args.steps):
            o2 = step2()
            n2 += sum(1 for h in o2 for s_ in h[1] if s_["status"] == 0 and s_["converged"])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        two_ctx = dict(scans_per_s=n2 / dt2, ms_per_step=dt2 * 1e3 / args.steps, over_one_context=(n2 / dt2) / (converged / elapsed),
                       note="the batch as two sub-batches of %d on two contexts (two HIP streams); same results" % (B // 2))
        c2[1].close()

    # ---- companion figure: NO step repeats a problem.  The timed steps above align the same 128 (scan, guess) problems every step,
    #      and the context keeps per-problem-index selection hints across calls (ProblemDev::qhint) -- a predictor no real stream
    #      offers.  Here slot b of step s holds scan (b + s) mod 64 with a guess drawn from a seed no other step uses: hints
    #      and scratch sizes come from a DIFFERENT problem, as in a live feed.  Never `value`.
    rotated = None
    if S == 1 and not args.no_fixed30 and not args.fixed_iters:
        ctx.set_params(check_every=args.check_every)

        def rot_problem(s_):
            rd = [d_scans[(b + s_) % len(d_scans)] for b in range(B)]
            ti = [w.T_truth[(b + s_) % len(d_scans)] @ synth.perturbation(77_000 + 1000 * rank + B * s_ + b) for b in range(B)]
            return rd, ti
        probs = [rot_problem(s_) for s_ in range(args.warmup + args.steps)]
        for rd, ti in probs[:args.warmup]:
            ctx.align_batch(map_id, rd, ti, raise_on_error=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nrot, irot = 0, []
        for rd, ti in probs[args.warmup:]:
            _, st_r = ctx.align_batch(map_id, rd, ti, raise_on_error=False)
            nrot += sum(1 for s_ in st_r if s_["status"] == 0 and s_["converged"])
            irot += [s_["iterations"] for s_ in st_r]
        torch.cuda.synchronize()
        dtr = time.perf_counter() - t0
        rotated = dict(scans_per_s=nrot / dtr, ms_per_step=dtr * 1e3 / args.steps, scans_converged=nrot, scans_total=args.steps * B,
                       mean_iterations=float(np.mean(irot)), over_repeated_batch=(nrot / dtr) / (converged / elapsed),
                       how="slot b of step s = scan (b + s) mod %d with a fresh guess (seed 77000 + B s + b): no problem is seen twice, "
                           "the cross-call selection hints come from a different problem" % len(d_scans))

    # ---- PCIe-inclusive companion figure: the caller owns HOST clouds (Localizer.hpp:103-126); step k+1's scans travel on
    #      the copy stream while step k aligns (pgicp_upload_f32), the compute stream waits for them on the device
    host_input = None
    if not args.no_host_input and S == 1:
        for c in ctxs:
            c.set_params(check_every=args.check_every)

        def host_loop(srcs, pinned):
            # the pipeline in its steady state: W untimed steps of the same pattern, then K timed ones; every step (the
            # last one too) uploads its successor's scans while it aligns
            cyc = [srcs[b % len(srcs)] for b in range(B)]
            cur = ctx.upload(cyc, pinned=pinned)
            ok = 0
            marks, up_ms = [], []
            for k in range(args.warmup + args.steps):
                if k == args.warmup:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ok = 0
                ta = time.perf_counter()
                nxt = ctx.upload(cyc, pinned=pinned)
                up_ms.append(round((time.perf_counter() - ta) * 1e3, 2))
                _, st_h = ctx.align_batch(map_id, cur, T_inits, raise_on_error=False)
                ok += sum(1 for s_ in st_h if s_["status"] == 0 and (s_["converged"] or args.fixed_iters))
                cur = nxt
                marks.append(round((time.perf_counter() - ta) * 1e3, 2))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            return ok / dt, dt * 1e3 / args.steps, dict(step_ms=marks, upload_call_ms=up_ms, untimed_first=args.warmup)

        # one pinned block holds the distinct scans back to back (a sensor driver's ring buffer): runs of equally spaced
        # scans travel as single 2-D transfers
        block = ctx.host_alloc((len(w.scans_xyz),) + w.scans_xyz[0].shape, np.float32)
        for q_, s_ in enumerate(w.scans_xyz):
            block[q_] = s_
        r_pin, ms_pin, each_pin = host_loop([block[q_] for q_ in range(len(w.scans_xyz))], True)
        r_page, ms_page, each_page = host_loop([np.ascontiguousarray(s_) for s_ in w.scans_xyz], False)
        ctx.host_free(block)
        dev_rate = converged / elapsed
        host_input = dict(pinned_scans_per_s=r_pin, pinned_ms_per_step=ms_pin, pinned_over_device_resident=r_pin / dev_rate,
                          pageable_scans_per_s=r_page, pageable_ms_per_step=ms_page, pageable_over_device_resident=r_page / dev_rate,
                          bytes_per_step=int(B * args.n_scan * 12), pinned_ms_each_step=each_pin, pageable_ms_each_step=each_page,
                          how="scans in host memory; step k+1 uploaded on the context's copy stream (pgicp_upload_f32) while step k "
                              "aligns; pageable sources pass through the context's pinned staging buffer (one host memcpy)")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, args.cpu_sample, os.cpu_count() or 1)
        cpu["gpu_over_single_core"] = ((converged_all / elapsed_max) / cpu["single_core_scans_per_s"]
                                       if cpu["single_core_scans_per_s"] else None)

    for c_, m_ in zip(ctxs, map_ids):
        c_.destroy_map(m_)
        c_.close()
    n_distinct = len(d_scans)
    del d_map_xyz, d_map_nrm, d_scans, readings
    torch.cuda.empty_cache()

    # ---- the other BASELINE configs under the same clock: compact legs, never `value` ----
    legs, lc_leg = workload_legs(args, world, rank)

    if rank == 0:
        value = converged_all / elapsed_max
        out = {
            "metric": ("ICP-converged scans/sec at 100k-pt scan vs 1M-pt local map" if not args.fixed_iters else
                       "scans/sec at 100k-pt scan vs 1M-pt local map, fixed 30 iterations"),
            "value": value,
            "unit": "scans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"scan-to-map ICP, {args.n_scan}-pt Velodyne-shaped scan vs {args.n_map}-pt local map, "
                                   f"<=30 iterations (BASELINE.json configs[1])",
                       "batch_scans_per_step": B, "distinct_scans": n_distinct, "matcher": args.matcher, "streams": S,
                       "chain": chain, "fixed_iterations": bool(args.fixed_iters),
                       "parallelism": f"{world} independent replica(s), one process per GPU"},
            "scans_total": scans_all,
            "scans_converged": converged_all,
            "fixed_30_iterations": fixed30,
            "two_contexts": two_ctx,
            "host_input": host_input,
            "rotated_batch": rotated,
            "mean_iterations": iters_all / max(1, scans_all),
            "selection_guess_misses_per_step": sel_fallbacks / max(1, args.steps + args.warmup),
            "set_map_ms": t_setmap * 1e3,
            "median_translation_error_m": float(np.median(err_t)) if err_t else None,
            "roofline": roofline,
            "kernels": kern,
            "cpu_baseline": cpu,
            "workloads": legs,
            "loop_closure": lc_leg,
            # the N > 1 evidence of the loop-closure leg, where the driver's parser sees it
            "rccl_ranks_seen": (lc_leg or {}).get("rccl_ranks_seen"),
            "comm_world_size": (lc_leg or {}).get("comm_world_size"),
            "edges_transport": (lc_leg or {}).get("edges_transport"),
            "pairs_per_s_one_gpu_same_run": (lc_leg or {}).get("pairs_per_s_one_gpu_same_run"),
            "speedup_vs_one_gpu": (lc_leg or {}).get("speedup_vs_one_gpu"),
            "launched_by": ("bench.py itself (launch_ranks)" if os.environ.get("PGSLAM_BENCH_SELF_LAUNCHED") else
                            ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process")),
            "command_wall_s": time.perf_counter() - t_cmd0,
        }
        if cpu and cpu["value"] > 0:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        # LAST key, compact (the driver keeps the tail of the line): every leg's figure and its roofline fraction under the
        # same clock -- [value, frac] per leg; the long form of each is in `workloads` above
        def vf(d_, key="value"):
            if not d_ or "error" in d_:
                return None
            r_ = d_.get("roofline") or {}
            return [round(float(d_[key]), 1), round(float(r_["frac"]), 4) if r_.get("frac") is not None else None]
        lg = legs or {}
        out["legs"] = {"unit": "[value, roofline.frac]; scans/s except loop_closure: pairs/s",
                       "headline": [round(value, 1), round(roofline["frac"], 4) if roofline else None],
                       "fixed30": [round(fixed30["scans_per_s"], 1), None] if fixed30 else None,
                       # what the boundary's callers get (Localizer.hpp:103-126 hands HOST clouds; no live feed repeats a problem)
                       "headline_host_pinned": [round(host_input["pinned_scans_per_s"], 1), None] if host_input else None,
                       "headline_host_pageable": [round(host_input["pageable_scans_per_s"], 1), None] if host_input else None,
                       "headline_rotated": [round(rotated["scans_per_s"], 1), None] if rotated else None,
                       "f64": vf(lg.get("f64")), "stream": vf(lg.get("stream")), "slam": vf(lg.get("slam")),
                       "slam_100k": vf(lg.get("slam_100k")),
                       # (the three-thread flavour of the same drive -- PoseGraphSlamMT, the reference's production flavour -- free running)
                       "slam_100k_mt": [round(float(((lg.get("slam_100k") or {}).get("slam_mt") or {}).get("scans_per_s")), 1), None]
                       if ((lg.get("slam_100k") or {}).get("slam_mt") or {}).get("scans_per_s") else None,
                       "loop_closure": vf(lg.get("loop_closure")),
                       "loop_closure_predicted_speedup_8": ((lg.get("loop_closure") or {}).get("shard_proxy") or {}).get("predicted_speedup", {}).get("8"),
                       "cpu_port_all_cores": round(cpu["value"], 2) if cpu else None}
        emit(out)
    dist_end()


if __name__ == "__main__":
    main()
