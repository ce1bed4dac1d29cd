#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (one directory per pass) per kernel:
average counter value per dispatch.  usage: tools/pmc_summary.py <prof_dir> [out.txt]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("k123_strip", "k1_vardct_group", "k1_scan", "k1_dct8", "k1_dct16_32", "k1_dct16", "k1_dct32", "k1_special", "k1_large_units", "k1_large_fused", "k1_large_llf", "k1_large_pass<1>", "k1_large_pass<2>", "k1_large", "k23_fused_filters", "k2_gaborish", "k3_epf", "k0b_lf_smooth", "k3_sigma_map",
                "k4_rct", "k5_palette", "k6_unsqueeze_flow", "k6_unsqueeze_rct", "k6_unsqueeze_levels", "k6_unsqueeze_tiled",
                "k6_unsqueeze"):
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
    # Derived per-kernel figures.  Normalisation (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / cycle constants):
    # SQ_* cycle counters are summed over the chip and count quad-cycles for the wave-level ones; GRBM_GUI_ACTIVE is
    # summed over the 8 XCDs.  VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8): the share
    # of the kernel's cycles in which a SIMD's vector ALU was issuing; LDS busy = SQ_LDS_IDX_ACTIVE / (256 CUs) over the
    # same denominator; the wait shares are fractions of SQ_WAVE_CYCLES (what a resident wavefront spends parked on
    # s_waitcnt / barriers, issue-stalled, or issuing).
    out.write("\nderived (see the comment in tools/pmc_summary.py for the normalisation)\n")
    for k in sorted(acc):
        a = {cn: sum(v) / len(v) for cn, v in acc[k].items()}
        need = ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_WAVES")
        if not all(n in a for n in need) or a["GRBM_GUI_ACTIVE"] == 0 or a["SQ_WAVE_CYCLES"] == 0:
            continue
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        out.write(f"  {k:44s} kernel cycles {cyc:10.0f}  VALU busy {a['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:5.1%}  "
                  f"LDS busy {a['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:5.1%} (conflict cycles {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):5.1%} of them)  "
                  f"wave cycles: parked {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:5.1%} issue-stalled {a['SQ_WAIT_INST_ANY'] / a['SQ_WAVE_CYCLES']:5.1%} "
                  f"issuing {a['SQ_ACTIVE_INST_ANY'] / a['SQ_WAVE_CYCLES']:5.1%}  VALU insts/wave {a['SQ_INSTS_VALU'] / max(a['SQ_WAVES'], 1):8.0f}\n")
    # HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled for wide
    # coalesced reads on gfx950, MI355X_MICROARCH.md section HBM)
    traffic = {}
    for k in acc:
        if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024
            w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024
            traffic[k] = {"fetch_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes": w,
                          "hbm_bytes": 2 * f + w}
    if traffic:
        import json
        out.write("\nHBM traffic per launch (bytes; fetch corrected x2)\n")
        for k, v in sorted(traffic.items()):
            out.write(f"  {k:40s} read {v['fetch_bytes_corrected'] / 1e6:10.1f} MB  write {v['write_bytes'] / 1e6:10.1f} MB\n")
        if len(sys.argv) > 3:
            json.dump(traffic, open(sys.argv[3], "w"), indent=1)
    stats = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
    if stats:
        out.write("\nkernel-trace stats (ns)\n")
        for row in csv.DictReader(open(stats[0])):
            out.write(f"  calls={row['Calls']:>5} avg_ns={float(row['AverageNs']):12.1f} pct={row['Percentage']:>6}  {row['Name'][:110]}\n")
    # the HIP-event numbers bench.py printed inside that same traced process (agreement check)
    log = os.path.join(d, "trace.log")
    if os.path.exists(log):
        import json
        import re
        m = re.search(r'\{"metric".*\}', open(log).read())
        if m:
            try:
                b = json.loads(m.group(0))
                k = b["roofline"]["all_kernels_ms_per_step"]
                out.write("\nHIP-event timing printed by bench.py inside the kernel-trace run (same process, same launches):\n")
                out.write("  " + "   ".join(f"{n} {v['ms_per_step']} ms" for n, v in k.items()) + f"   step {b['ms_per_step']} ms\n")
            except Exception as e:  # noqa: BLE001 - a malformed log must not break the summary
                out.write(f"\n(bench line in trace.log not parseable: {e})\n")


if __name__ == "__main__":
    main()
