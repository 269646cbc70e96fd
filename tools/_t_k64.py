import sys, os, statistics
sys.path[:0] = ["/root/repo", "/root/repo/nextgen-uia_amd"]
import torch
from uia_hip import ops
dev = torch.device("cuda", 0)
M, N, K = 50432, 768, 64
a = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
pw = ops.PackedW(w)
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
outs = {}
for cfg in (0, 14, 23, 8):
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(a, pw if cfg != 23 else w, out_t=o, tile_cfg=cfg)
    try:
        t = statistics.median(timeit(f) for _ in range(3))
        outs[cfg] = o.clone()
        print("cfg", cfg, round(t, 1), "us", "equal to auto:", torch.equal(outs[cfg], outs[0]))
    except Exception as e:
        print("cfg", cfg, "error", str(e)[:200])
