"""The ring GEMM on the square shapes the programming guide quotes its 256² 8-phase template on (4096³ / 8192³ bf16, random [-1, 1) operands, store-only epilogue):
is this library's K loop at that level once K is long enough for prologue and epilogue not to matter?  Also the step's own K = 768 shapes at the same M x N for contrast.
GPU box:  python tools/gemm_square_yardstick.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops, functional as UF

dev = torch.device("cuda", 0)
def timeit(f, n=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
CFGS = tuple(int(c) for c in os.environ.get("YARD_CFGS", "27,28,25,24,12").split(","))
SHAPES = ((4096, 4096, 4096), (8192, 8192, 8192), (65536, 768, 3072), (65536, 2304, 768), (65536, 3072, 768), (65536, 768, 768), (65536, 4096, 4096))
if os.environ.get("YARD_SHAPES"):
    SHAPES = tuple(tuple(int(v) for v in s.split("x")) for s in os.environ["YARD_SHAPES"].split(","))
for M, N, K in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    a_rows = a
    if os.environ.get("YARD_KB"):
        a = ops.KBlocked(a_rows.view(M, K // 32, 32).permute(1, 0, 2).contiguous())       # the step's own activation layout: [K/32][M][32]
    w = ops.PackedW((torch.rand(N, K, device=dev) * 2 - 1).bfloat16())
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    try:
        t = timeit(lambda: ops.gemm(a, w, out_t=out))
    except Exception:                                       # small M with a K-blocked A: the automatic choice is not a ring config
        t = timeit(lambda: ops.gemm(a, w, out_t=out, tile_cfg=8))
    want = out.clone()
    extra = ""
    for cfg in CFGS:
        try:
            out.zero_()
            ops.gemm(a, w, out_t=out, tile_cfg=cfg)
            same = bool(torch.equal(out, want))
            tc = timeit(lambda: ops.gemm(a, w, out_t=out, tile_cfg=cfg))
            extra += f"  cfg{cfg} {2.0 * M * N * K / tc * 1e-12:7.1f}{'' if same else ' (DIFFERS)'}"
        except Exception as e:
            extra += f"  cfg{cfg} n/a ({str(e)[:40]})"
    wr = w._row
    t2 = timeit(lambda: torch.matmul(a_rows, wr.t()))
    fl = 2.0 * M * N * K
    print(f"M {M:6d} N {N:5d} K {K:5d}: uia_gemm {t * 1e6:8.1f} us  {fl / t * 1e-12:7.1f} TF/s ({fl / t * 1e-12 / 2500:.3f} of 2.5 PF)   torch.matmul (hipBLASLt) {t2 * 1e6:8.1f} us  {fl / t2 * 1e-12:7.1f} TF/s |{extra}", flush=True)
