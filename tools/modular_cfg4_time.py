"""BASELINE config 4 pieces (bench.py's time_modular_config without the CPU leg): chain + RCT, palette, RCT alone, ms.
usage: JXLH_LIBRARY=<so> python tools/modular_cfg4_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import jxl_rs_amd  # noqa: E402
import bench  # noqa: E402

r = bench.time_modular_config(jxl_rs_amd, np, 0, 8192, steps=8, cores=1, cpu=False)
print(json.dumps({"chain_ms": r["chain"]["ms"], "palette_ms": r["palette"]["ms"], "rct_ms": r["rct_alone"]["ms"]}))
