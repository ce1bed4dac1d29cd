#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2 rocpd sqlite) result into the text summary kept under profiles/.
usage: tools/rocprof_summary.py <results.db> [<out.txt>] [--title "..."]"""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    title = ""
    if "--title" in sys.argv:
        title = sys.argv[sys.argv.index("--title") + 1]
        args = [a for a in args if a != title]
    db = args[0]
    out = open(args[1], "w") if len(args) > 1 else sys.stdout
    c = sqlite3.connect(db)
    if title:
        out.write(title + "\n")
    out.write("rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
    out.write(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel\n")
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        out.write(f"{calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name}\n")
    try:
        out.write("\nper-kernel launch geometry / resources (first dispatch of each kernel)\n")
        seen = set()
        for row in c.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, "
                             "sgpr_count, scratch_size from kernels order by start"):
            if row[0] in seen:
                continue
            seen.add(row[0])
            out.write(f"  grid=({row[1]},{row[2]}) wg={row[3]} lds={row[4]} vgpr={row[5]} agpr={row[6]} sgpr={row[7]} "
                      f"scratch={row[8]}  {row[0]}\n")
    except sqlite3.Error as e:  # schema drift between ROCm versions
        out.write(f"(kernel table unavailable: {e})\n")


if __name__ == "__main__":
    main()
