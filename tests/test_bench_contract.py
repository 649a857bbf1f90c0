"""The bench line's contract, checked on the round's recorded run (`python bench.py --steps 20 --warmup 5` on the GPU box):
profiles/rNN_bench_n1.json is the LINE bench.py printed (round 6 on: the short form of bench.compact_line), and
profiles/rNN_bench_n1_full.json the full record it wrote to bench_full.json (up to round 5 the line WAS the full record).
Checked: the fields the driver reads, the roofline and cpu_baseline objects and their arithmetic, and that the printed line is
short enough for the driver to keep whole (BENCH_r05.json: `parsed: null` -- the line had grown to 22.9 KB)."""
import glob, json, os, sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def latest():
    """(full record, BASELINE.json)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_n1.json")))
    if not files:
        pytest.skip("no recorded bench line")
    full = files[-1].replace("_bench_n1.json", "_bench_n1_full.json")
    return json.load(open(full if os.path.exists(full) else files[-1])), json.load(open(os.path.join(ROOT, "BASELINE.json")))


def strict_loads(text):
    def refuse(name):
        raise ValueError("not strict JSON: " + name)
    return json.loads(text, parse_constant=refuse)


def test_printed_line_is_short_strict_json_with_roofline_and_cpu_baseline():
    """emit()'s own serialisation of the latest full record: < 6000 characters, strict JSON (no NaN / Infinity), the contract's keys,
    flat `roofline` / `cpu_baseline` objects, `legs` last."""
    import bench
    full, _ = latest()
    line = json.dumps(bench.compact_line(full), allow_nan=False)
    assert len(line) < bench.LINE_MAX <= 6000, len(line)
    d = strict_loads(line)
    assert list(d)[-1] == "legs"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    r, c = d["roofline"], d["cpu_baseline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches", "active_problems_per_launch",
              "algorithmic_bytes_per_launch", "bound_measured"):
        assert k in r, k
    assert all(not isinstance(v, (dict, list)) for v in r.values())          # no per-launch arrays, no nested evidence
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert d["legs"]["headline"][0] == pytest.approx(d["value"], rel=1e-3)
    # a record that outgrows the line sheds optional facts, never the contract
    fat = dict(full, launched_by="x" * 20000)
    assert len(json.dumps(bench.compact_line(fat))) < bench.LINE_MAX


def test_recorded_printed_line_is_the_short_form():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_n1.json")))
    if not files or not os.path.exists(files[-1].replace("_bench_n1.json", "_bench_n1_full.json")):
        pytest.skip("no round-6 record yet")
    text = open(files[-1]).read().strip()
    assert len(text) < 6000 and "\n" not in text
    d = strict_loads(text)
    assert list(d)[-1] == "legs" and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    # what the boundary's callers get: host clouds (pinned / pageable), and a batch no step repeats
    for leg in ("headline_host_pinned", "headline_host_pageable", "headline_rotated"):
        assert d["legs"][leg][0] > 0, leg
    assert d["legs"]["headline_rotated"][0] > 0.5 * d["value"]


def test_recorded_line_keeps_the_contract():
    d, base = latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "scans/s" and isinstance(d["value"], float) and d["value"] > 0
    # value = converged scans of all steps / the timed region; ms_per_step the same region per step
    batch = d["config"]["batch_scans_per_step"]
    assert d["value"] == pytest.approx(d["scans_converged"] / (d["steps"] * d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["scans_total"] == d["steps"] * batch
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # achieved = algorithmic bytes per launch / the kernel's average launch duration (HIP events on the context's stream)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9, rel=1e-6)
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["bound_measured"] == "valu" and 0.0 < r["frac_unseeded_launches"] < r["frac"] < r["frac_seeded_launches"] < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    # the companion legs: never `value`, each with its own roofline
    for leg in ("loop_closure", "stream", "slam", "f64", "slam_100k"):
        w = d["workloads"][leg]
        assert "error" not in w, (leg, w)
        assert w["value"] > 0 and "roofline" in w and 0.0 < w["roofline"]["frac"] < 1.0
    assert d["workloads"]["f64"]["value"] >= 0.45 * d["value"]           # the double path at about half the float rate (0.48-0.51 by run)
    assert len(d["workloads"]["stream"]["scans_per_s_each_pass"]) >= 3
    if "legs" in d:                                                      # (round 5 on)
        # the LAST key, compact: the driver keeps the tail of the line -- every leg's [value, roofline.frac] must be in it
        assert list(d.keys())[-1] == "legs" and len(json.dumps(d["legs"])) < 700
        for leg in ("headline", "f64", "stream", "slam", "slam_100k", "loop_closure"):
            v = d["legs"][leg]
            assert v[0] > 0 and 0.0 < v[1] < 1.0
        assert d["legs"]["headline"][0] == pytest.approx(d["value"], rel=1e-3)
        assert "bound_means" in r
        # the one-GPU proxy of the 8-GPU target (north_star: >= 3.5x on batched loop closing)
        sp = d["workloads"]["loop_closure"]["shard_proxy"]
        assert d["legs"]["loop_closure_predicted_speedup_8"] == pytest.approx(sp["predicted_speedup"]["8"])
        assert sp["predicted_speedup"]["8"] >= 3.5 and len(sp["per_world"]["8"]["shard_ms"]) == 8
        assert sp["predicted_speedup"]["2"] < sp["predicted_speedup"]["4"] < sp["predicted_speedup"]["8"]
        # the SLAM legs time passes AFTER a warm one, in one process
        assert d["workloads"]["slam_100k"]["slam"]["passes"] >= 4 and len(d["workloads"]["slam_100k"]["slam"]["pass_slam_s"]) >= 4


def test_metric_is_baselines():
    d, base = latest()
    assert base["metric"].split()[0].lower() in d["metric"].lower() or "scans/sec" in d["metric"]
