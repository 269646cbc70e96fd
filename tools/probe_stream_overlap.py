"""Probe: do memory-bound row kernels (LayerNorm) of one HIP stream run UNDER the MFMA-bound K loops of a GEMM stream on the same GPU?
(GEMM workgroup: 8 waves x 188 VGPRs + 128 KB LDS per CU -> room for ~2 more waves/SIMD of <= 64 VGPRs and 32 KB LDS.)
Prints serial vs concurrent wall time for n GEMMs on stream A and m LayerNorms on stream B."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

dev = torch.device("cuda", 0)
M = 50432
a = torch.randn(M, 768, device=dev).bfloat16()
w = ops.PackedW((torch.randn(2304, 768, device=dev) * 0.03).bfloat16())
bias = torch.randn(2304, device=dev)
out = torch.empty(M, 2304, device=dev, dtype=torch.bfloat16)
x = torch.randn(M, 768, device=dev)
g, b = torch.ones(768, device=dev), torch.zeros(768, device=dev)
y = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
x2 = torch.randn(M, 768, device=dev); r2 = torch.randn(M, 768, device=dev); o2 = torch.empty(M, 768, device=dev)
a2 = torch.randn(M, 768, device=dev).bfloat16(); w2 = ops.PackedW((torch.randn(768, 768, device=dev) * 0.03).bfloat16()); b2 = torch.randn(768, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def gemms(n, kind):
    for _ in range(n):
        if kind == "qkv":
            ops.gemm(a, w, bias=bias, out_t=out, tile_cfg=8)
        else:
            ops.gemm(a2, w2, bias=b2, resid=r2, out32=o2, tile_cfg=8)


def lns(m):
    for _ in range(m):
        ops.layernorm_fwd(x, g, b, 1e-6, y_t=y)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for kind in ("qkv", "proj_resid32"):
    n, m = 60, 180
    gemms(5, kind); lns(5)
    tg = timed(lambda: gemms(n, kind))
    tl = timed(lambda: lns(m))

    def both():
        with torch.cuda.stream(sa):
            gemms(n, kind)
        with torch.cuda.stream(sb):
            lns(m)
    tb = timed(both)

    def two_gemm_streams():
        with torch.cuda.stream(sa):
            gemms(n // 2, kind)
        with torch.cuda.stream(sb):
            gemms(n // 2, kind)
    t2 = timed(two_gemm_streams)
    print(f"{kind}: {n} GEMMs alone {tg:.2f} ms | {m} LayerNorms alone {tl:.2f} ms | serial sum {tg + tl:.2f} | two streams {tb:.2f} ms "
          f"(hidden {100 * (tg + tl - tb) / min(tg, tl):.0f} % of the shorter) | same GEMMs split over two streams {t2:.2f} ms")
