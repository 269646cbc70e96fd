"""How much of the text tower's N = 768 launches (epilogue mask 977: deferred-LayerNorm fp32 residual in, fp32 + K-blocked bf16 out, row sums) is the
fp32 residual traffic?  Times, at M = 65 536, the same GEMM with fewer epilogue bytes per element: 10 B (today) / 6 B (bf16 residual in, bf16 + fp32 out
... ) / 4 B (bf16 in, bf16 out) / 2 B (plain).  An upper bound for what a 3-byte residual format could buy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M, N = 65536, 768
dev = "cuda"
for K in (768, 3072):
    a = torch.randn(M, K, device=dev).bfloat16()
    g = ops.kb_group(torch.bfloat16)
    a_kb = ops.KBlocked(a.view(M, K // g, g).permute(1, 0, 2).contiguous())
    w = ops.PackedW((torch.randn(N, K, device=dev) * K ** -0.5).bfloat16())
    bias = torch.randn(N, device=dev)
    r32 = torch.randn(M, N, device=dev)
    rT = r32.bfloat16()
    o32 = torch.empty(M, N, device=dev)
    oT = ops.kb_empty(M, N, torch.bfloat16, dev)
    sums = torch.zeros(M, 2, device=dev, dtype=torch.int64)
    stats = torch.stack([r32.mean(1), r32.var(1, unbiased=False).add(1e-12).rsqrt()], 1).contiguous()
    lw, lb = torch.randn(N, device=dev), torch.randn(N, device=dev)
    res = {}
    res["10 B: deferred-LN fp32 resid, fp32 + bf16 out, row sums (mask 977)"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid=r32, resid_ln=(stats, lw, lb), out32=o32, out_t=oT, rowsum=sums))
    hi3, lo3 = ops.float_to_three_byte(r32)
    hi3 = ops.KBlocked(hi3.view(M, N // g, g).permute(1, 0, 2).contiguous())
    olo = torch.empty(M, N, device=dev, dtype=torch.int8)
    res[" 6 B: THREE-BYTE deferred-LN resid in, hi + lo out, row sums; low bytes row-major"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid3=(hi3, lo3), resid_ln=(stats, lw, lb), out_t=oT, out_lo=olo, rowsum=sums))
    lo3b = ops.KBlocked(lo3.view(M, N // 64, 64).permute(1, 0, 2).contiguous())
    olob = ops.kb_empty(M, N, torch.int8, dev)
    res[" 6 B: THREE-BYTE ..., low bytes in 64-column blocks (round 4)"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid3=(hi3, lo3b), resid_ln=(stats, lw, lb), out_t=oT, out_lo=olob, rowsum=sums))
    res["10 B: fp32 resid, fp32 + bf16 out, row sums"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid=r32, out32=o32, out_t=oT, rowsum=sums))
    res[" 8 B: fp32 resid, fp32 out (mask 81)"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid=r32, out32=o32))
    res[" 6 B: bf16 resid, fp32 out"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid_t=rT, out32=o32))
    res[" 4 B: bf16 resid, bf16 out"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, resid_t=rT, out_t=oT))
    res[" 2 B: bias, bf16 out"] = timed(lambda: ops.gemm(a_kb, w, bias=bias, out_t=oT))
    print(f"M {M} N {N} K {K}")
    for k, v in res.items():
        print(f"   {k:75s} {v:7.1f} us")
