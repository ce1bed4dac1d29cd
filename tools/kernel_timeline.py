#!/usr/bin/env python3
"""Start / end / gap of consecutive kernel dispatches of a rocprofv3 rocpd database (one timed step).
usage: tools/kernel_timeline.py <results.db> [first_kernel_substring] [n]"""
import sqlite3
import sys

db = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k1_scan"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if first in r[0]]
i0 = starts[len(starts) // 2] if starts else 0
prev_end = None
for name, s, e in rows[i0:i0 + n]:
    gap = "" if prev_end is None else f"{(s - prev_end) / 1000:8.1f}"
    short = name.split("(")[0].split("::")[-1][:40]
    print(f"{short:42s} dur {(e - s) / 1000:9.1f} us  gap {gap}")
    prev_end = e
