"""jxlh_rct on three 8192^2 planes, ms per call (JXLH_LIBRARY selects the build)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import jxl_rs_amd
from jxl_rs_amd.lib import DeviceArray
n = 8192 * 8192
c = jxl_rs_amd.Context(0, 1)
d = [DeviceArray(nbytes=n * 4) for _ in range(3)]
res = []
for rep in range(3):
    c._chk(c.L.jxlh_rct(c._ctx, d[0].ptr, d[1].ptr, d[2].ptr, n, 6, 0), "rct")
    c.sync()
    c.timer_start()
    for _ in range(10):
        c._chk(c.L.jxlh_rct(c._ctx, d[0].ptr, d[1].ptr, d[2].ptr, n, 6, 0), "rct")
    res.append(round(c.timer_stop() / 10, 4))
print(json.dumps({"rct_ms": res, "GBs": round(24.0 * n / (min(res) * 1e-3) / 1e9, 1)}))
