"""Times uia_gemm on the GEMM shapes of one config-2 training step (ViT-B/16 + Mona at bs 256, BERT-base at 256 x 256 tokens), one
epilogue mask each as the step uses it, under the host-side scheduling variants of uia_hip.ops.gemm.

    python tools/gemm_shapes_bench.py [--iters 20] [--variants base,kb,split,kb+split] [--cfg N]

Prints per shape: launches/step, microseconds and TFLOP/s per variant, and the step-weighted total.  Run on the GPU box."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from uia_hip import ops

MV, MT = 256 * 197, 256 * 256
# (label, launches per step, M, N, K, epilogue)
SHAPES = [
    ("vit qkv", 12, MV, 2304, 768, "bias"),
    ("vit proj", 12, MV, 768, 768, "resid32"),
    ("vit fc1+stash", 11, MV, 3072, 768, "gelu_stash"),
    ("vit fc2", 12, MV, 768, 3072, "resid32"),
    ("mona project1", 12, MV, 64, 768, "bias"),
    ("mona project2", 12, MV, 768, 64, "resid32"),
    ("vit fc2 dgrad", 11, MV, 3072, 768, "dgelu"),
    ("vit fc1 dgrad", 11, MV, 768, 3072, "plain"),
    ("vit proj dgrad", 11, MV, 768, 768, "plain"),
    ("vit qkv dgrad", 11, MV, 768, 2304, "plain"),
    ("mona p2 dgrad", 12, MV, 64, 768, "plain"),
    ("mona p1 dgrad", 12, MV, 768, 64, "plain"),
    ("bert qkv", 12, MT, 2304, 768, "bias"),
    ("bert proj", 12, MT, 768, 768, "resid32"),
    ("bert fc1", 12, MT, 3072, 768, "gelu"),
    ("bert fc2", 12, MT, 768, 3072, "resid32"),
]


def run(shape, variant, iters, cfg):
    _, _, M, N, K, epi = shape
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16
    a = torch.randn(M, K, device=dev).to(dt)
    w = ops.PackedW((torch.randn(N, K, device=dev) * K ** -0.5).to(dt))
    bias = torch.randn(N, device=dev)
    kw = {}
    if epi == "resid32":
        kw = dict(bias=bias, resid=torch.randn(M, N, device=dev), out32=torch.empty(M, N, device=dev))
    elif epi == "bias":
        kw = dict(bias=bias, out_t=torch.empty(M, N, device=dev, dtype=dt))
    elif epi == "plain":
        kw = dict(out_t=torch.empty(M, N, device=dev, dtype=dt))
    elif epi == "gelu":
        kw = dict(bias=bias, act="gelu", out_t=torch.empty(M, N, device=dev, dtype=dt))
    elif epi == "gelu_stash":
        kw = dict(bias=bias, act="gelu", aux_out=torch.empty(M, N, device=dev, dtype=dt), out_t=torch.empty(M, N, device=dev, dtype=dt))
    elif epi == "dgelu":
        kw = dict(dact="gelu", aux_in=torch.randn(M, N, device=dev).to(dt), out_t=torch.empty(M, N, device=dev, dtype=dt))
    ops.KBLOCK_W, ops.TAIL_SPLIT = "kb" in variant, "split" in variant
    try:
        for _ in range(3):
            ops.gemm(a, w, tile_cfg=cfg, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(a, w, tile_cfg=cfg, **kw)
        e1.record()
        torch.cuda.synchronize()
    finally:
        ops.KBLOCK_W, ops.TAIL_SPLIT = True, True
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variants", default="base,kb,split,kb+split")
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    variants = args.variants.split(",")
    totals = {v: 0.0 for v in variants}
    print(f"{'shape':16s} {'x':>3s} {'M':>6s} {'N':>5s} {'K':>5s} {'epi':>10s} | " + " | ".join(f"{v:>16s}" for v in variants))
    for sh in SHAPES:
        if args.only and args.only not in sh[0]:
            continue
        name, n, M, N, K, epi = sh
        cells = []
        for v in variants:
            us = run(sh, v, args.iters, args.cfg)
            totals[v] += us * n
            cells.append(f"{us:7.1f}us {2.0 * M * N * K / us * 1e-6:6.0f}TF")
        print(f"{name:16s} {n:3d} {M:6d} {N:5d} {K:5d} {epi:>10s} | " + " | ".join(cells), flush=True)
    print("step total (ms): " + "  ".join(f"{v}={totals[v] * 1e-3:.2f}" for v in variants))


if __name__ == "__main__":
    main()
