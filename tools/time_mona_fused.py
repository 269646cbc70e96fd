"""Times the Mona adapter forward at the ViT-B/16 shape (256 images x 197 tokens x 768, bf16): uia_mona_fused_fwd against the four unfused
launches, over three rotating input / output sets (465 MB of x: past the Infinity Cache, as inside the step).  GPU box: python tools/time_mona_fused.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops
B, h, w, D = 256, 14, 14, 768
N, M = 1 + h * w, 256 * 197
dev, dt = "cuda", torch.bfloat16
NS = 3
xs = [torch.randn(B, N, D, device=dev) for _ in range(NS)]
ys = [torch.empty(B, N, D, device=dev) for _ in range(NS)]
yts = [ops.kb_empty(M, D, dt, dev) for _ in range(NS)]
sums = [torch.zeros(M, 2, device=dev, dtype=torch.int64) for _ in range(NS)]
u, t, d = torch.empty(M, D, device=dev, dtype=dt), torch.empty(M, 64, device=dev, dtype=dt), torch.empty(M, 64, device=dev, dtype=dt)
nw, nb, g, gx = (torch.randn(D, device=dev) for _ in range(4))
w1 = ops.PackedW((torch.randn(64, D, device=dev) * 0.03).to(dt)); b1 = torch.randn(64, device=dev) * 0.1
w2 = ops.PackedW((torch.randn(D, 64, device=dev) * 0.05).to(dt)); b2 = torch.randn(D, device=dev) * 0.1
shapes = dict(conv1_w=(64, 9), conv1_b=(64,), conv2_w=(64, 25), conv2_b=(64,), conv3_w=(64, 49), conv3_b=(64,), proj_w=(64, 64), proj_b=(64,), freq=(64,))
P = {k: torch.randn(*s, device=dev) * 0.1 for k, s in shapes.items()}


def fused(i):
    ops.mona_fused_fwd("freq_enhanced", B, h, w, xs[i], nw, nb, g, gx, w1.row, b1, w2.row, b2, P, ys[i], y_t=yts[i], rowsum=sums[i], u_out=u, t_out=t, d_out=d, p_drop=0.1, seed=5)


def fused_nostash(i):
    ops.mona_fused_fwd("freq_enhanced", B, h, w, xs[i], nw, nb, g, gx, w1.row, b1, w2.row, b2, P, ys[i], y_t=yts[i], rowsum=sums[i], t_out=t, d_out=d, p_drop=0.1, seed=5)


def unfused(i):
    ops.mona_pre_fwd(xs[i], nw, nb, g, gx, u)
    ops.gemm(u, w1, bias=b1, out_t=t)
    ops.mona_spatial_fwd("freq_enhanced", B, h, w, t, P, d, p_drop=0.1, seed=5)
    sums[i].zero_()
    ops.gemm(d, w2, bias=b2, resid=xs[i].view(M, D), out32=ys[i].view(M, D), out_t=yts[i], rowsum=sums[i])


def timeit(f, n=12):
    for i in range(3):
        f(i % NS)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i % NS)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


v = os.environ.get("MF_V", "99")
line = f"variant {v}: fused {timeit(fused):7.1f} us | fused without the u stash {timeit(fused_nostash):7.1f} us"
if v == "99":
    line += f" | four unfused launches (+ one fill) {timeit(unfused):7.1f} us"
print(line, flush=True)
