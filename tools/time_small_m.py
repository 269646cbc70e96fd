"""Which tile config for small-M / narrow-N bf16 GEMMs (CLIPSeg's prompt tower at M = 77, its decoder's N = 64 reductions)?  Run on the GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timeit(fn, n=40):
    for i in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in ((77, 512, 2048), (77, 2048, 512), (77, 512, 512), (77, 1536, 512), (128, 1024, 1024), (128, 4096, 1024), (128, 1024, 4096), (1280, 3072, 768),
                (1, 512, 512), (25216, 64, 2048), (25216, 64, 768), (25216, 2048, 64)):
    a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(torch.bfloat16))
    bias = torch.randn(N, generator=g).to(dev)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for cfg in (0, 3, 4, 5, 13, 14, 8):
        try:
            res[cfg] = timeit(lambda: ops.gemm(a, w, bias=bias, out_t=o, tile_cfg=cfg))
        except Exception as e:
            res[cfg] = float("nan")
    print(f"M={M} N={N} K={K}: " + "  ".join(f"cfg{c} {v:.1f}" for c, v in res.items()), flush=True)
