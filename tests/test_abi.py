"""CPU-side checks of the drop-in boundary: libpgicp.so loads and exports every
symbol include/pgicp.h declares; no compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from pgslam_amd import icp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pgicp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgicp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared() == sorted(icp.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = icp.load_library()
    for name in _declared():
        assert hasattr(lib, name), name


def test_abi_version_and_struct_sizes():
    lib = icp.load_library()
    assert lib.pgicp_abi_version() == 6
    assert ctypes.sizeof(icp.Edge) == 512
    p = icp.Params()
    lib.pgicp_default_params(ctypes.byref(p))
    # libpointmatcher defaults of the chain pgslam instantiates
    assert (p.knn, p.epsilon, p.trim_ratio, p.max_iters, p.smooth_length) == (1, 0.0, 0.85, 40, 3)
    assert p.max_dist == float("inf") and p.min_diff_rot == 0.001 and p.min_diff_trans == 0.001


def test_no_cpu_fallback_without_gpu():
    lib = icp.load_library()
    if lib.pgicp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(icp.PgicpError) as e:
        icp.Context(0)
    assert e.value.code == icp.ERR_NO_DEVICE


def test_shard_pairs_lpt_is_a_partition():
    costs = np.array([9, 1, 8, 2, 7, 3, 6, 4, 5, 5], dtype=np.int64)
    parts = [icp.shard_pairs(costs, 4, r) for r in range(4)]
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(10))
    loads = [int(costs[p].sum()) for p in parts]
    assert max(loads) - min(loads) <= int(costs.max())
    # deterministic
    assert all(np.array_equal(icp.shard_pairs(costs, 4, r), parts[r]) for r in range(4))
    # uniform costs deal evenly
    sizes = [len(icp.shard_pairs(np.ones(512, dtype=np.int64), 8, r)) for r in range(8)]
    assert sizes == [64] * 8


def test_check_icp_result_follows_loop_closer():
    ok = dict(status=0, max_iter_reached=False, overlap=0.9)
    assert icp.check_icp_result(ok, 100.0)
    assert not icp.check_icp_result(dict(ok, max_iter_reached=True), 100.0)      # LoopCloser.hpp:317
    assert not icp.check_icp_result(dict(ok, overlap=0.79), 100.0)               # LoopCloser.hpp:331
    assert not icp.check_icp_result(ok, 5000.1)                                  # LoopCloser.hpp:335
    assert icp.check_icp_result(ok, 5000.0)
    assert not icp.check_icp_result(dict(ok, status=1), 1.0)
    # the record-array form (what a batch of loop-closure candidates uses): the same function on every record
    rng = np.random.default_rng(3)
    sa = np.zeros(64, dtype=icp._STATS_DTYPE)
    sa["status"] = rng.integers(0, 2, 64) * rng.integers(0, 2, 64)
    sa["max_iter_reached"] = rng.integers(0, 2, 64)
    sa["overlap"] = rng.uniform(0.6, 1.0, 64)
    res = rng.uniform(0.0, 10000.0, 64)
    got = icp.check_icp_results(sa, res, 0.8, 5000.0)
    want = [icp.check_icp_result(dict(status=int(r["status"]), max_iter_reached=bool(r["max_iter_reached"]), overlap=float(r["overlap"])), float(x), 0.8, 5000.0)
            for r, x in zip(sa, res)]
    assert got.tolist() == [int(w) for w in want] and 0 < got.sum() < 64


def test_numpy_views_of_the_batch_records_match_the_ctypes_structures():
    """Context.align_batch fills pgicp_problem[] and reads pgicp_stats[] through numpy structured views: field for field the
    layout must be the C structures' (include/pgicp.h, mirrored by the ctypes classes)."""
    import ctypes as C
    from pgslam_amd import icp
    assert icp._PROBLEM_DTYPE.itemsize == C.sizeof(icp.Problem)
    assert icp._STATS_DTYPE.itemsize == C.sizeof(icp.Stats)
    for name in icp._PROBLEM_DTYPE.names:
        assert icp._PROBLEM_DTYPE.fields[name][1] == getattr(icp.Problem, name).offset, name
    for name in icp._STATS_DTYPE.names:
        assert icp._STATS_DTYPE.fields[name][1] == getattr(icp.Stats, name).offset, name
    assert set(icp._PROBLEM_DTYPE.names) == {f[0] for f in icp.Problem._fields_}
    assert set(icp._STATS_DTYPE.names) == {f[0] for f in icp.Stats._fields_}


def test_header_is_plain_c99_and_its_records_are_the_bindings(tmp_path):
    """include/pgicp.h is what a cgo / JNI / ctypes binding reads: it must compile as strict C99, link from C, and its records
    must have the sizes the ctypes mirror in pgslam_amd/icp.py assumes (no compute: pgicp_device_count answers 0 without a GPU)."""
    import ctypes as C
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "pgicp.h"\n#include <stdio.h>\nint main(void)\n{\n    pgicp_params p;\n    pgicp_default_params(&p);\n'
                   '    printf("%d %d %zu %zu %zu %zu %zu\\n", pgicp_abi_version(), p.knn, sizeof(pgicp_params), sizeof(pgicp_stats), sizeof(pgicp_problem),\n'
                   '           sizeof(pgicp_edge), sizeof(pgicp_filter));\n    return pgicp_device_count() >= 0 ? 0 : 1;\n}\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "pgslam_amd", "lib")
    exe = str(tmp_path / "abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"), str(src), "-o", exe,
                           "-L" + lib, "-lpgicp", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    assert out.returncode == 0, out.stderr
    abi, knn, s_params, s_stats, s_problem, s_edge, s_filter = (int(v) for v in out.stdout.split())
    assert abi == 6 and knn == 1
    assert (s_params, s_stats, s_problem, s_edge, s_filter) == (C.sizeof(icp.Params), C.sizeof(icp.Stats), C.sizeof(icp.Problem), C.sizeof(icp.Edge),
                                                                C.sizeof(icp.Filter))
