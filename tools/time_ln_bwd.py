"""LayerNorm backward, image-tower shape (50 432 x 768): the plain form (fp32 residual gradient in and out + bf16 copy: 16 B per element) against three-byte
residual gradients in and out (12 B per element).  GPU box: python tools/time_ln_bwd.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
from uia_hip import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, D = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 197, int(sys.argv[2]) if len(sys.argv) > 2 else 768


def timeit(f, it=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / it * 1e3, 1)


torch.manual_seed(0)
dy = torch.randn(M, D, device=dev).bfloat16()
x = torch.randn(M, D, device=dev)
gamma = torch.rand(D, device=dev) + 0.5
dres = torch.randn(M, D, device=dev)
dx32, dx_t = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev, dtype=torch.bfloat16)
r_hi, r_lo = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(M, D, device=dev, dtype=torch.int8)
o_hi, o_lo = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(M, D, device=dev, dtype=torch.int8)
# a three-byte image of dres: LayerNorm backward of zero dy with dres as the residual writes exactly round3(dres)
ops.layernorm_bwd(torch.zeros_like(dy), x, gamma, 1e-5, dres=dres, dx_t=r_hi, dx_lo=r_lo)
plain = lambda: ops.layernorm_bwd(dy, x, gamma, 1e-5, dres=dres, dx32=dx32, dx_t=dx_t)
three = lambda: ops.layernorm_bwd(dy, x, gamma, 1e-5, dres=(r_hi, r_lo), dx_t=o_hi, dx_lo=o_lo)
plain(); three()
back = (o_hi.view(torch.int16).int() << 16 | (o_lo.int() << 8)).view(torch.float32) if False else None
val3 = ((o_hi.view(torch.int16).to(torch.int32) << 16) + (o_lo.to(torch.int32) << 8)).view(torch.float32)
print("three-byte vs plain, max rel err:", float((val3 - dx32).abs().max() / dx32.abs().max()))
GB = M * D / 1e9
for name, f, b in (("plain 16 B/el", plain, 16), ("three-byte 12 B/el", three, 12)):
    t = timeit(f)
    print(f"{name}: {t} us  {GB * b / t * 1e3:.2f} TB/s", flush=True)
