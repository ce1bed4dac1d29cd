#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (one directory per pass) per kernel:
average counter value per dispatch.  usage: tools/pmc_summary.py <prof_dir> [out.txt]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("k1_vardct_group", "k23_fused_filters", "k2_gaborish", "k3_epf", "k0b_lf_smooth", "k3_sigma_map",
                "k4_rct", "k5_palette", "k6_unsqueeze"):
        if key in name:
            return key + (name[name.index("<"):name.index(">") + 1] if "<" in name and key in ("k3_epf", "k23_fused_filters") else "")
    return None


def main():
    d = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k is None:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out.write("rocprofv3 --pmc: mean counter value per kernel dispatch (separate passes per counter set)\n")
    for k in sorted(acc):
        out.write(f"\n{k}\n")
        for cn in sorted(acc[k]):
            v = acc[k][cn]
            out.write(f"  {cn:28s} {sum(v) / len(v):18.1f}   (n={len(v)})\n")
    stats = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
    if stats:
        out.write("\nkernel-trace stats (ns)\n")
        for row in csv.DictReader(open(stats[0])):
            out.write(f"  calls={row['Calls']:>5} avg_ns={float(row['AverageNs']):12.1f} pct={row['Percentage']:>6}  {row['Name'][:110]}\n")


if __name__ == "__main__":
    main()
