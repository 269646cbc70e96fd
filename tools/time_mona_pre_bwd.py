"""Times uia_mona_pre_bwd (+ its reduction) at the ViT-B/16 shape (50432 x 768, bf16 du).  GPU box: python tools/time_mona_pre_bwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops
M, D = 256 * 197, 768
dev = "cuda"
du = torch.randn(M, D, device=dev).bfloat16(); x = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev)
nw, nb, g, gx = (torch.randn(D, device=dev) for _ in range(4))
dx, dxt = torch.empty_like(x), torch.empty_like(du)
G = [torch.zeros(D, device=dev) for _ in range(4)]
f = lambda: ops.mona_pre_bwd(du, x, dy, nw, nb, g, gx, dx, dxt, *G)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
nbytes = M * D * (2 + 4 + 4 + 4 + 2)
print(f"mona_pre_bwd + reduce: {us:.1f} us, {nbytes / us * 1e-6:.2f} TB/s algorithmic")
