"""Timing of the K = 64 rank-update launches (tile cfg 23 against cfg 14) at the ViT-L/14 + LoRA shape.  Run on the GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

dev = torch.device("cuda", 0)
M, N = 32896, 1024
g = torch.Generator().manual_seed(0)
t = torch.randn(M, 64, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, 64, generator=g) * 0.1).to(dev).to(torch.bfloat16)
bufs = [torch.randn(M, N, generator=g).to(dev).to(torch.bfloat16) for _ in range(6)]     # 6 x 67 MB: rotate so that the Infinity Cache does not hold the operand
outs = [torch.empty_like(b) for b in bufs]


def timeit(fn, n=30):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cfg in (14, 23):
    r = {}
    r["in_place"] = timeit(lambda i: ops.gemm(t, w, alpha=2.0, resid_t=bufs[i % 6], out_t=bufs[i % 6], tile_cfg=cfg))
    r["out_of_place"] = timeit(lambda i: ops.gemm(t, w, alpha=2.0, resid_t=bufs[i % 6], out_t=outs[i % 6], tile_cfg=cfg))
    r["in_place_drop"] = timeit(lambda i: ops.gemm(t, w, alpha=2.0, resid_t=bufs[i % 6], out_t=bufs[i % 6], tile_cfg=cfg, drop=("acc", 0.1, 77)))
    r["same_buffer"] = timeit(lambda i: ops.gemm(t, w, alpha=2.0, resid_t=bufs[0], out_t=bufs[0], tile_cfg=cfg))
    print(f"cfg {cfg}: " + "  ".join(f"{k} {v:.1f} us" for k, v in r.items()), flush=True)
x = torch.randn(M, 1024, generator=g).to(dev).to(torch.bfloat16)
a = (torch.randn(64, 1024, generator=g) * 0.03).to(dev).to(torch.bfloat16)
tt = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
xd = torch.empty_like(x)
print("skinny64: plain %.1f us  drop %.1f us  drop+byproduct %.1f us" % (
    timeit(lambda i: ops.gemm(bufs[i % 6], a, out_t=tt)),
    timeit(lambda i: ops.gemm(bufs[i % 6], a, out_t=tt, drop=("a", 0.1, 5))),
    timeit(lambda i: ops.gemm(bufs[i % 6], a, out_t=tt, drop=("a", 0.1, 5, outs[i % 6])))), flush=True)
