"""Where is the bf16 image tower's feature error made?  An emulation, in fp32 torch ops on the GPU, of the roundings the
bf16 path applies — every GEMM operand and every stored activation rounded to bf16, accumulation and the residual stream in fp32 — with each
rounding SITE switchable, on the benchmark's model (ViT-B/16 + 12 Mona freq_enhanced, adapters scaled away from their init as in
tools/parity_at_bench_batch.py).  Prints, against the all-fp32 evaluation of the same oracle code: the error with every site on (compare
with the kernels' measured figure), with one site on, and with one site off.  tools/ only; the product path is not involved.

    python tools/parity_error_budget.py [--batch 32] [--stress]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
import torch.nn.functional as F
from oracle import vit_ref, mona_ref

SITES = ["A_qkv", "W_qkv", "O_qkv", "attn_P", "O_attn", "W_proj", "A_fc1", "W_fc1", "O_gelu", "W_fc2", "mona_u", "W_p1", "mona_t", "mona_d", "W_p2"]
ON = set()
rb = lambda t: t.bfloat16().float()
EXACT_W = set()          # data_ptr of frozen weights kept UNROUNDED (--exact-weight-blocks: what a bf16 hi + bf16 lo split of those weights would give, to 2^-16)


def R(site, t):
    if site not in ON or (site.startswith("W_") and t.data_ptr() in EXACT_W):
        return t
    return rb(t)


_lin, _gelu, _attn = F.linear, F.gelu, vit_ref._attention
KIND = {(2304, 768): "qkv", (768, 768): "proj", (3072, 768): "fc1", (768, 3072): "fc2", (64, 768): "p1", (768, 64): "p2"}
IN_MONA = [False]


def linear(x, w, b=None):
    k = KIND.get(tuple(w.shape))
    if k == "qkv":
        return R("O_qkv", _lin(R("A_qkv", x), R("W_qkv", w), b))
    if k == "proj":
        return _lin(x, R("W_proj", w), b)                 # A = attention output (site O_attn); result joins the fp32 residual
    if k == "fc1":
        return _lin(R("A_fc1", x), R("W_fc1", w), b)
    if k == "fc2":
        return _lin(x, R("W_fc2", w), b)                  # A = gelu output (site O_gelu)
    if k == "p1":
        return R("mona_t", _lin(R("mona_u", x), R("W_p1", w), b))
    if k == "p2":
        return _lin(x, R("W_p2", w), b)                   # A = d (site mona_d)
    return _lin(x, w, b)


def gelu(x):
    y = _gelu(x)
    return R("mona_d", y) if x.shape[-1] == 64 else R("O_gelu", y)


def attention(q, k, v, heads, mask=None):
    B, L, D = q.shape
    dh = D // heads
    q = q.view(B, L, heads, dh).transpose(1, 2)
    k = k.view(B, L, heads, dh).transpose(1, 2)
    v = v.view(B, L, heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2) * dh ** -0.5
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    o = (R("attn_P", p) @ v) / p.sum(-1, keepdim=True)   # the kernel rounds the un-normalised exponentials, sums them in fp32
    return R("O_attn", o.transpose(1, 2).reshape(B, L, D))


def features(images, P):
    F.linear, F.gelu, vit_ref._attention = linear, gelu, attention
    try:
        with torch.no_grad():
            return vit_ref.timm_vit_forward(images, P, heads=12, mona=dict(variant="freq_enhanced", hw=(14, 14)))
    finally:
        F.linear, F.gelu, vit_ref._attention = _lin, _gelu, _attn


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--stress", action="store_true")
    ap.add_argument("--exact-weight-blocks", default="", help="round 6 (VERDICT r05 item 5): comma-separated lists of blocks whose frozen weights stay unrounded, ';' between "
                    "experiments, e.g. '0,1,10,11;0,1,2,3;8,9,10,11' — prints max-norm / rms of the image features with every other site on, then exits")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("pabb", os.path.join(ROOT, "tools", "parity_at_bench_batch.py"))
    pabb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pabb)
    from src.adapters import inject_mona_variant_to_open_clip
    from src.third_party.biomedclip.model import create_biomedclip
    import contextlib, io
    g = torch.Generator().manual_seed(41 + args.batch)
    model = create_biomedclip(seed=3)
    with contextlib.redirect_stdout(io.StringIO()):
        inject_mona_variant_to_open_clip(model, variant="freq_enhanced", bottleneck_dim=64)
    pabb.scale_adapters(model, g)
    if args.stress:
        pabb.stress_weights(model)
    dev = torch.device("cuda", 0)
    P = {k: v.detach().to(dev) for k, v in model.state_dict().items() if k.startswith("visual.")}
    images = torch.rand(args.batch, 3, 224, 224, generator=g).to(dev)
    ON.clear()
    ref = features(images, P)
    ON.update(SITES)
    e_all = rel(features(images, P), ref)
    print(f"batch {args.batch}{' (outlier stress)' if args.stress else ''}: every site on: image features rel {e_all:.2e}   (rms {float((features(images, P) - ref).pow(2).mean().sqrt() / ref.abs().max()):.2e})")
    if args.exact_weight_blocks:
        import re
        for exp in args.exact_weight_blocks.split(";"):
            blocks = {int(b) for b in exp.split(",") if b.strip() != ""}
            EXACT_W.clear()
            for k, v in P.items():
                m = re.match(r"visual\.trunk\.blocks\.(\d+)\.(attn|mlp)\..*weight$", k)
                if m and int(m.group(1)) in blocks and v.dim() == 2:
                    EXACT_W.add(v.data_ptr())
            f = features(images, P)
            print(f"frozen weights of blocks {sorted(blocks)} unrounded ({len(EXACT_W)} matrices), every other site on: max-norm {rel(f, ref):.2e}   rms {float((f - ref).pow(2).mean().sqrt() / ref.abs().max()):.2e}")
        EXACT_W.clear()
        return
    print(f"{'site':10s} {'only this site':>16s} {'all but this site':>18s}")
    groups = [("A_qkv",), ("A_fc1",), ("W_qkv", "W_proj", "W_fc1", "W_fc2"), ("O_qkv",), ("attn_P",), ("O_attn",), ("O_gelu",), ("mona_u",), ("mona_t",), ("mona_d",), ("W_p1", "W_p2")]
    for grp in groups:
        ON.clear(); ON.update(grp)
        e1 = rel(features(images, P), ref)
        ON.clear(); ON.update(s for s in SITES if s not in grp)
        e2 = rel(features(images, P), ref)
        print(f"{'+'.join(grp)[:28]:28s} {e1:10.2e} {e2:14.2e}")
    for name, grp in (("whole Mona adapters", ("mona_u", "W_p1", "mona_t", "mona_d", "W_p2")), ("attention (qkv out, P, O)", ("O_qkv", "attn_P", "O_attn")),
                      ("MLP (A_fc1, gelu out, weights)", ("A_fc1", "O_gelu", "W_fc1", "W_fc2"))):
        ON.clear(); ON.update(grp)
        e1 = rel(features(images, P), ref)
        ON.clear(); ON.update(s for s in SITES if s not in grp)
        print(f"{name:28s} {e1:10.2e} {rel(features(images, P), ref):14.2e}")


if __name__ == "__main__":
    main()
