import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import jxl_rs_amd
n = 8192
ctx = jxl_rs_amd.Context(0, 1)
dev = "cuda:0"
out = torch.empty((n, n), dtype=torch.int32, device=dev)
a2 = torch.randint(0, 256, (n // 2, n // 2), dtype=torch.int32, device=dev)
ah = torch.randint(0, 256, (n, n // 2), dtype=torch.int32, device=dev)
av = torch.randint(0, 256, (n // 2, n), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
ctx.kernel_timing(True)
for rep in range(6):
    ctx.smooth_unsqueeze_dev(2, a2, n // 2, n // 2, n // 2, out, n, n, n)
    ctx.smooth_unsqueeze_dev(0, ah, n // 2, n // 2, n, out, n, n, n)
    ctx.smooth_unsqueeze_dev(1, av, n, n, n // 2, out, n, n, n)
ctx.sync()
for k, (ms, cnt) in ctx.kernel_times().items():
    print(k, round(ms / cnt, 4), cnt)
