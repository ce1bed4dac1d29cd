#!/usr/bin/env python3
"""Merged kernel + memory-copy timeline from a rocprofv3 rocpd database (--kernel-trace --memory-copy-trace).
usage: tools/timeline_dump.py <results.db> <t0_ms> <t1_ms>   (window relative to the first kernel)"""
import sqlite3
import sys

db, w0, w1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
c = sqlite3.connect(db)
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
for name, s, e, q in c.execute("select name, start, end, queue_id from kernels"):
    ev.append((s, e, "K q%s %s" % (q, name.split("(anonymous namespace)::")[-1].split("(")[0][:36])))
mc = [n for n in names if "memory_cop" in n and not n.startswith("rocpd_")]
if mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({mc[0]})")]
    size_col = "size" if "size" in cols else None
    for row in c.execute(f"select name, start, end{', size' if size_col else ''} from {mc[0]}"):
        ev.append((row[1], row[2], "M %s %s" % (row[0][:24], (str(round(row[3] / 1e6, 2)) + " MB") if size_col else "")))
else:
    print("no memory copy table among", names[:40])
ev.sort()
t00 = min(e[0] for e in ev if e[2].startswith("K"))
for s, e, what in ev:
    a, b = (s - t00) / 1e6, (e - t00) / 1e6
    if a >= w0 and a <= w1:
        print(f"{a:9.3f} {b:9.3f} {b - a:8.3f}  {what}")
