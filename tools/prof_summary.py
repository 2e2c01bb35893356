#!/usr/bin/env python3
"""Filter a rocprofv3 --kernel-trace --stats CSV down to this repo's kernels and write a compact summary
(the bench harness's torch data-generation kernels are dropped).  Usage: prof_summary.py kernel_stats.csv out.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mine = [r for r in rows if any(k in r["Name"] for k in ("k1_", "k12_", "k_fused", "k2a_", "k2b_", "k2w_", "k2c_", "k2d_", "k2u_", "k2_classes", "k_compact", "k_sort", "k_reverse", "k_literal", "k_join", "k_flag", "k_records", "k_identity", "k_copy_records"))]
tot = sum(float(r["TotalDurationNs"]) for r in mine) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "share_of_pipeline_pct"])
    for r in sorted(mine, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].split("(")[0].replace("void ", "")
        w.writerow([name, r["Calls"], "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3),
                    "%.1f" % (float(r["TotalDurationNs"]) / 1e3), "%.1f" % (100 * float(r["TotalDurationNs"]) / tot)])
print(open(sys.argv[2]).read())
