"""Times uia_layernorm_fwd / _bwd at the step's shapes (ViT-B: 50432 x 768, bf16 operand out; BERT: 65536 x 768, bf16 operand + row
statistics, and the same with the fp32 output).  GPU box: python tools/time_layernorm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


D = 768
g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
for name, M in (("vit", 50432), ("bert", 65536)):
    x = torch.randn(M, D, device="cuda")
    yt = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty_like(x)
    st = torch.empty(M, 2, device="cuda")
    t = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-5, y_t=yt))
    print(f"{name} fwd  bf16 out         : {t:6.1f} us  {M * D * 6 / t * 1e-6:5.2f} TB/s")
    t = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-5, y_t=yt, stats=st))
    print(f"{name} fwd  bf16 out + stats : {t:6.1f} us  {M * D * 6 / t * 1e-6:5.2f} TB/s")
    t = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-5, y_t=yt, y32=y32))
    print(f"{name} fwd  bf16 + fp32 out  : {t:6.1f} us  {M * D * 10 / t * 1e-6:5.2f} TB/s")
    dy = torch.randn(M, D, device="cuda").bfloat16()
    dres = torch.randn(M, D, device="cuda")
    dx, dxt = torch.empty_like(x), torch.empty_like(dy)
    t = timed(lambda: ops.layernorm_bwd(dy, x, g, 1e-5, dres=dres, dx32=dx, dx_t=dxt))
    print(f"{name} bwd                   : {t:6.1f} us  {M * D * 16 / t * 1e-6:5.2f} TB/s")
