#!/usr/bin/env python3
"""What binds k_knn_grid, from rocprofv3 --pmc passes of the headline command (tools/pmc_round.sh): per dispatch and over
the step, VALU busy = 4 x SQ_INSTS_VALU / (1 024 SIMDs x cycles) (a wave64 VALU instruction holds a 16-lane SIMD for four
cycles), texture-address busy = TA_TA_BUSY_sum / (256 CUs x cycles), with cycles = GRBM_GUI_ACTIVE / 8 (the counter sums
over the XCDs).  Writes profiles/knn_pmc.json, which bench.py quotes in roofline.bound_measured_evidence.

  pmc_valu.py DIR_GRBM_TA DIR_SQ_INSTS OUT.json QUERIES_PER_LAUNCH [KERNEL [MEAN_ITERATIONS_PER_QUERY]]"""
import csv, glob, json, os, sys, collections
d_grbm, d_sq, out, queries = sys.argv[1:5]
kernel = sys.argv[5] if len(sys.argv) > 5 else "k_knn_grid"
mean_iters = float(sys.argv[6]) if len(sys.argv) > 6 else None     # iterations a query takes part in on average (the bench line's mean_iterations)
def per_dispatch(d):
    per = collections.OrderedDict()
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]
g, s = per_dispatch(d_grbm), per_dispatch(d_sq)
n = min(len(g), len(s))
rows = []
for i in range(n):
    cyc = g[i]["GRBM_GUI_ACTIVE"] / 8.0
    valu = s[i]["SQ_INSTS_VALU"]
    rows.append(dict(launch=i, cycles_per_xcd=cyc, valu_busy=4.0 * valu / (1024.0 * cyc),
                     ta_busy=g[i].get("TA_TA_BUSY_sum", 0.0) / (256.0 * cyc),
                     valu_wave_insts=valu, vmem_rd_wave_insts=s[i].get("SQ_INSTS_VMEM_RD"), salu_wave_insts=s[i].get("SQ_INSTS_SALU"),
                     lds_wave_insts=s[i].get("SQ_INSTS_LDS")))
def busy(sel):
    c = sum(r["cycles_per_xcd"] for r in sel); v = sum(r["valu_wave_insts"] for r in sel)
    return 4.0 * v / (1024.0 * c) if c else None
q = float(queries)
# the first launch of a step has no correspondences to start from; a step is the launches up to the next long one
first = [r for i, r in enumerate(rows) if i == 0 or r["cycles_per_xcd"] > 2.0 * rows[i - 1]["cycles_per_xcd"]]
rest = [r for r in rows if r not in first]
tot_valu = sum(r["valu_wave_insts"] for r in rows)
res = dict(kernel=kernel, commit=os.environ.get("PGSLAM_COMMIT"), bound_measured="valu", launches=n, steps_in_record=len(first),
           valu_busy_all_launches=busy(rows), valu_busy_unseeded_launches=busy(first), valu_busy_seeded_launches=busy(rest),
           ta_busy_all_launches=sum(r["ta_busy"] * r["cycles_per_xcd"] for r in rows) / sum(r["cycles_per_xcd"] for r in rows),
           valu_lane_ops_per_query_iteration=64.0 * tot_valu / (q * n),
           valu_lane_ops_per_active_query_iteration=(64.0 * tot_valu / (q * mean_iters * max(1, len(first)))) if mean_iters else None,
           valu_wave_insts_per_step=tot_valu / max(1, len(first)),
           vmem_rd_wave_insts_per_step=sum(r["vmem_rd_wave_insts"] or 0 for r in rows) / max(1, len(first)),
           queries_per_launch_nominal=q,
           per_launch=rows,
           how="rocprofv3 --pmc, two separate passes (GRBM_GUI_ACTIVE TA_TA_BUSY_sum | SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS) "
               "of `bench.py --steps 1 --warmup 0` without the companion legs; VALU busy = 4 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
           note="valu_lane_ops_per_query_iteration counts every query of the batch in every launch; converged problems leave the launch: "
                "valu_lane_ops_per_active_query_iteration divides by queries x the mean iterations per query instead")
json.dump(res, open(out, "w"), indent=1)
print({k: v for k, v in res.items() if k != "per_launch"})
