"""cfg 8 (4-deep ring) against cfg 24 (5-deep, 160 KB of LDS) on the step's large shapes, plain bf16 epilogue, K-blocked W, rotating A buffers.  Run on the GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timeit(fn, n=20):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in ((65536, 2304, 768), (65536, 3072, 768), (65536, 768, 3072), (65536, 768, 768), (43520, 3072, 768), (43520, 768, 3072), (43520, 2304, 768)):
    A = [torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16) for _ in range(3)]
    w = ops.PackedW((torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(torch.bfloat16))
    bias = torch.randn(N, generator=g).to(dev)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    r = {}
    for rep in range(2):
        for cfg in (8, 24):
            r.setdefault(cfg, []).append(timeit(lambda i: ops.gemm(A[i % 3], w, bias=bias, out_t=o, tile_cfg=cfg)))
    print(f"M={M} N={N} K={K}: " + "  ".join(f"cfg{c} {min(v):.1f} us" for c, v in r.items()), flush=True)
    del A, o
