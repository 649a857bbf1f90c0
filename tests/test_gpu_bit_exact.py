"""A whole ICP, bit for bit.  Up to round 5 T_out / cov / residual could only be compared "to 1e-12": the device added the pairs'
terms in its sorting order in a tree of its own, the oracle in scan order in one chain, and double addition is not associative.
Round 6 states ONE reduction tree (oracle/icp_oracle.c "RT-1"; k_p2plane_reduce / k_cov_reduce /
block_reduce_store / sum_partials_256 walk it) and lets the pairs enter it in ONE agreed order:
  * sum_order = SCAN: the caller's reading order -- the oracle's default; results depend on the inputs alone;
  * sum_order = SORTED (the default, all loads coalesced): the library's sorting order of the reading, which the test reads back
    (pgicp_debug_reading_order) and hands to the oracle as `pair_order`: a permutation is all that is borrowed.
Either way every iteration's 27 sums, hence every T_iter, every match of every later iteration, the covariance and the
residual must come out IDENTICAL -- np.array_equal on the doubles.  What the sums stand for in the reference: the error
minimiser inside ICP::operator() (Localizer.hpp:126, LoopCloser.hpp:98), getCovariance (Localizer.hpp:238), getResidualError
(LoopCloser.hpp:362)."""
import numpy as np
import pytest

from pgslam_amd import icp, synth

pytestmark = pytest.mark.gpu

CHAIN = dict(max_dist=2.0, trim_ratio=0.85, max_iters=30, min_diff_rot=0.001, min_diff_trans=0.01,
             smooth_length=3, sensor_std_dev=0.01)


def same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def check(st, T, o, what):
    assert st["status"] == 0 and o["status"] == 0, what
    assert st["iterations"] == o["iterations"] and st["converged"] == o["converged"], what
    assert same_bits(T, o["T"]), (what, np.abs(T - o["T"]).max())
    assert same_bits(st["cov"], o["cov"]), (what, "cov")
    assert same_bits(st["residual"], o["residual"]) and same_bits(st["overlap"], o["overlap"]), (what, st["residual"], o["residual"])
    assert st["n_kept"] == o["n_kept"] and same_bits(st["trim_limit"], o["trim_limit"]), what


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("sum_order", [icp.SUM_ORDER_SORTED, icp.SUM_ORDER_SCAN])
def test_two_scans_whole_icp_bit_for_bit(oracle32, oracle64, dtype, sum_order):
    orc = oracle32 if dtype == np.float32 else oracle64
    s = synth.make_two_scans(10_000, rings=16)                       # BASELINE configs[0]
    rd, ref, nrm = (s[k].astype(dtype) for k in ("reading_xyz", "ref_xyz", "ref_nrm"))
    ctx = icp.Context(0, **CHAIN, sum_order=sum_order)
    mid = ctx.set_map(ref, nrm, center=True, dtype=dtype)
    for seed in range(4):
        T0 = s["T_init"] @ synth.perturbation(300 + seed)
        T, st = ctx.align(mid, rd, T0, dtype=dtype)
        order = ctx.reading_order(len(rd)) if sum_order == icp.SUM_ORDER_SORTED else None
        assert order is None or sorted(order.tolist()) == list(range(len(rd)))
        o = orc.icp(rd, ref, nrm, T0, pair_order=order, **CHAIN)
        check(st, T, o, (np.dtype(dtype).name, sum_order, seed))
    ctx.destroy_map(mid)
    ctx.close()


def test_batch_of_ragged_problems_bit_for_bit(oracle32):
    """several problems of different sizes in one device batch (one reduce launch for all of them): each problem's order is its own"""
    ctx = icp.Context(0, **CHAIN)
    w = synth.make_scan_to_map(n_scan=6000, n_map=40_000, n_queries=3, n_map_poses=3, rings=16)
    mid = ctx.set_map(w.map_xyz, w.map_nrm, center=True)
    readings = [w.scans_xyz[0], w.scans_xyz[1][:4321], w.scans_xyz[2][:2049], w.scans_xyz[0][:2048], w.scans_xyz[1][:777]]
    T0 = [w.T_init[0], w.T_init[1], w.T_init[2], w.T_init[0], w.T_init[1]]
    for sum_order in (icp.SUM_ORDER_SORTED, icp.SUM_ORDER_SCAN):
        ctx.set_params(sum_order=sum_order)
        T, st = ctx.align_batch(mid, readings, T0)
        for p, rd in enumerate(readings):
            order = ctx.reading_order(len(rd), problem=p) if sum_order == icp.SUM_ORDER_SORTED else None
            o = oracle32.icp(rd, w.map_xyz, w.map_nrm, T0[p], pair_order=order, **CHAIN)
            check(st[p], T[p], o, (sum_order, p))
    ctx.destroy_map(mid)
    ctx.close()


@pytest.mark.parametrize("variant", ["p2point", "knn3", "median_maxdist", "robust_cauchy"])
def test_other_chains_bit_for_bit(oracle32, variant):
    """the other modules whose sums pass through the tree: Kabsch sums (PointToPoint), knn > 1 (pairs [point][neighbour]), another
    quantile filter, robust weights inside the sums"""
    over = dict(p2point=dict(error_minimizer=1), knn3=dict(knn=3), median_maxdist=dict(trim_ratio=0.5, quantile_scale=3.0, outlier_max_dist=0.8),
                robust_cauchy=dict(trim_ratio=1.0, robust_fct=1, robust_tuning=1.0, robust_scale=1))[variant]
    chain = dict(CHAIN, **over)
    s = synth.make_two_scans(6000, rings=16)
    for sum_order in (icp.SUM_ORDER_SORTED, icp.SUM_ORDER_SCAN):
        ctx = icp.Context(0, **chain, sum_order=sum_order)
        mid = ctx.set_map(s["ref_xyz"], s["ref_nrm"], center=True)
        T, st = ctx.align(mid, s["reading_xyz"], s["T_init"])
        order = ctx.reading_order(len(s["reading_xyz"])) if sum_order == icp.SUM_ORDER_SORTED else None
        o = oracle32.icp(s["reading_xyz"], s["ref_xyz"], s["ref_nrm"], s["T_init"], pair_order=order, **chain)
        if variant == "robust_cauchy":                         # (welsch / cauchy weights are T-typed transcendental-free here: cauchy is a division)
            assert st["iterations"] == o["iterations"]
        check(st, T, o, (variant, sum_order))
        ctx.destroy_map(mid)
        ctx.close()


def test_stage_level_error_stats_equal_the_oracles_sums(oracle32):
    """pgicp_error_stats takes the caller's ids in reading order: its tree positions ARE scan positions"""
    s = synth.make_two_scans(9000, rings=16)
    ctx = icp.Context(0, **CHAIN)
    mid = ctx.set_map(s["ref_xyz"], s["ref_nrm"], center=True)
    rd = oracle32.transform(s["T_init"], s["reading_xyz"])
    ids, d2 = ctx.match(mid, rd)
    w, _, _ = ctx.outlier_weights(d2)
    wratio, residual, dsys = ctx.error_stats(mid, rd, ids, w)
    mean = oracle32.centroid(s["ref_xyz"])
    st, sys_ = oracle32.p2plane_system(rd - mean, s["ref_xyz"] - mean, s["ref_nrm"], ids, w)
    assert st == 0
    assert same_bits(dsys, sys_), np.abs(dsys - sys_).max()
    assert same_bits(residual, sys_[29]) and same_bits(wratio, sys_[27] / len(rd))
    ctx.destroy_map(mid)
    ctx.close()
