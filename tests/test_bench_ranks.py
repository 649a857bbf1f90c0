"""bench.py's N > 1 path on a box without devices: `bench.py --gpus 2` started by hand (launch_ranks: the parent starts two
fresh rank processes) runs the loop-closure rank skeleton with the library's HOST transport (pgicp_comm_create_host) and a
stand-in aligner (PGSLAM_BENCH_DRY_RANKS=1, bench.main_loopclosure_dry) and must print ONE short line that carries the
multi-rank evidence SURVEY.md 8(e) asks for -- and no line at all, with a non-zero exit code, when a rank's edges are missing.
What is sharded: /root/reference/src/pgslam/LoopCloserMT.hpp:45-67 (one ICP at a time)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(world, **env):
    e = dict(os.environ, PGSLAM_BENCH_DRY_RANKS="1", **env)
    e.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", "loopclosure", "--pairs", "37",
                           "--steps", "3", "--warmup", "0"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("world", [2, 3])
def test_launch_ranks_line_carries_the_multi_rank_evidence(world):
    p = run(world)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines[-1]) < 6000
    d = json.loads(lines[-1])
    assert d["dry_run"] is True and d["n_gpus"] == world
    assert d["rccl_ranks_seen"] == d["comm_world_size"] == world
    assert d["pairs_per_s_one_gpu_same_run"] > 0 and d["speedup_vs_one_gpu"] > 0
    assert d["speedup_vs_one_gpu"] == pytest.approx(d["value"] / d["pairs_per_s_one_gpu_same_run"], rel=1e-4)
    assert list(d)[-1] == "legs"


def test_a_rank_whose_edges_are_missing_fails_the_run():
    p = run(2, PGSLAM_BENCH_DRY_DROP_RANK="1")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "not a 2-GPU measurement" in p.stderr
