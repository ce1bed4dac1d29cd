import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import jxl_rs_amd
from jxl_rs_amd import lib
c = jxl_rs_amd.Context(0, 1)
for n in (1024, 4096, 8192):
    rng = np.random.default_rng(1)
    idx = torch.from_numpy(rng.integers(0, 40, size=(n, n)).astype(np.int32)).cuda()
    pal = torch.from_numpy(rng.integers(-10, 256, size=(3, 64)).astype(np.int32)).cuda()
    out = torch.empty((3, n, n), dtype=torch.int32, device="cuda")
    def run():
        c._chk(c.L.jxlh_palette_delta(c._ctx, lib._addr(idx), n, n, lib._addr(pal), 56, 8, 64, 3, 8, 5, lib._addr(out)), "pd")
    run(); c.sync()
    t0 = time.perf_counter(); run(); c.sync(); t = time.perf_counter() - t0
    print(n, "delta palette gradient predictor: %.2f ms  (%.1f MP/s)" % (t * 1e3, n * n / t / 1e6))
