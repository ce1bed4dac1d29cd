"""Where a kernel's scratch (spill) accesses come from: compiles one .hip with line tables and maps every
scratch_load / scratch_store of the chosen kernel to its source line.
usage: python tools/spill_map.py <file.hip> <kernel-symbol-substring> [extra hipcc flags...]"""
import re
import subprocess
import sys
from collections import Counter
import os

src, sym = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
out = "/tmp/spill_map.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                "-gline-tables-only", "-I" + os.path.dirname(os.path.abspath(src)), "-S", "--cuda-device-only", "-o", out,
                src] + extra, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
files = {}
for l in lines[:end]:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur = None
st, ld = Counter(), Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
    if "scratch_store" in l:
        st[cur] += 1
    if "scratch_load" in l:
        ld[cur] += 1
print("kernel", lines[start].split(":")[0], "instructions", sum(1 for l in lines[start:end] if l.startswith("\t") and not l.lstrip().startswith((".", ";"))))
for name, c in (("stores", st), ("loads", ld)):
    print(name, sum(c.values()))
    for k, v in sorted(c.items(), key=lambda kv: (-kv[1])):
        print("  ", k, v)
