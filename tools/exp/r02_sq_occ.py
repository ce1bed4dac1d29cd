import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import jxl_rs_amd
from jxl_rs_amd import lib
ctx = jxl_rs_amd.Context(0, 1)
L = ctx.L
dev = "cuda:0"
n = 8192
for lines in (4096, 8192, 16384, 24576, 32768):
    avg = torch.randint(0, 256, (lines, n // 2), dtype=torch.int32, device=dev)
    res = torch.randint(-8, 9, (lines, n // 2), dtype=torch.int32, device=dev)
    out = torch.empty((lines, n), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    P = lib._addr
    f = lambda: ctx._chk(L.jxlh_unsqueeze(ctx._ctx, 1, P(avg), n // 2, P(res), n // 2, n, lines, P(out), n), "u")
    f(); ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        f()
    ms = ctx.timer_stop() / 5
    print(lines, "lines (", lines // 64, "WGs ) H:", round(ms, 4), "ms")
    del avg, res, out
