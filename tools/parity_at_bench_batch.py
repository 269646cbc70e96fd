"""Parity evidence at the size the benchmark runs.  Run once on the GPU box; not part of `pytest -m gpu`.

The configs[1] step — ViT-B/16 + 12 Mona (freq_enhanced) + BERT-base, InfoNCE at tau = 0.07 — at B = 64 and B = 256, bf16 mode, every
optimisation of the bench path active (ring GEMMs, folded LayerNorms, K-blocked activations), against oracle/train_ref.py's arithmetic on
the host cores: image / text features, loss, whole-gradient cosine / relative L2, median and worst per-tensor gradient error.

The oracle cannot hold the autograd graph of 256 images at once (≈ 0.2 GB per image in fp32), so it runs in two passes that compute
EXACTLY the same gradient: (1) features of every chunk without a graph, the loss and d loss / d image-features on the full batch;
(2) per chunk: forward with a graph, backward of <features, d loss / d features>, gradients summed over chunks.

Stress case (--stress, B = 64): outlier channels and non-centred residual rows — pos_embed / cls_token get one channel at 30 sigma and a
row offset of 2 sigma, the BERT LayerNorm biases likewise — to show whether the 1e-2 bound holds or the LayerNorm-fold guard trips.

    python tools/parity_at_bench_batch.py --batches 64,256 [--stress] [--out gpurun_out/parity_bench_batch.json]
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from oracle import losses_ref, text_ref, vit_ref


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def captions(g, B, L=256):
    ids = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(24, 129, (1,), generator=g))
        ids[b, 1:n - 1] = torch.randint(1000, 30000, (n - 2,), generator=g)
        ids[b, 0], ids[b, n - 1] = 2, 3
    return ids


def scale_adapters(model, gen, std=0.03):
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "mona" not in k.lower():
                continue
            if k.endswith(("norm.weight", "gammax", "freq_filter")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif k.endswith("gamma"):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
            elif p.dim() >= 2:
                p.copy_(std * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=gen))


def stress_weights(model):
    """Outlier channels and a non-zero row mean in BOTH residual streams."""
    with torch.no_grad():
        tr = model.visual.trunk
        s = float(tr.pos_embed.std())
        tr.pos_embed[..., 5] += 30.0 * s                    # one channel at 30 sigma in every token row of every block
        tr.pos_embed += 2.0 * s                             # row mean = 2 sigma
        tr.cls_token[..., 5] += 30.0 * s
        for k, p in model.named_parameters():
            if k.startswith("text.") and k.endswith("LayerNorm.bias"):       # post-LN: the stream is re-normalised, outliers live in the LN bias
                p[7] += 3.0
                p += 0.2


def oracle_step(P, trainable, images, ids, variant, chunk, threads):
    torch.set_num_threads(threads)
    mona = dict(variant=variant, hw=(14, 14))
    B = images.shape[0]
    t0 = time.perf_counter()
    with torch.no_grad():
        fi = torch.cat([vit_ref.timm_vit_forward(images[i:i + chunk], P, heads=12, mona=mona) for i in range(0, B, chunk)])
        ft = torch.cat([text_ref.bert_text_forward(ids[i:i + chunk], P, heads=12) for i in range(0, B, chunk)])
    fi_leaf = fi.clone().requires_grad_(True)
    loss = losses_ref.info_nce(fi_leaf, ft, 0.07)
    loss.backward()
    dfi = fi_leaf.grad
    leaves = {k: P[k].detach().clone().requires_grad_(True) for k in trainable}
    Pq = dict(P)
    Pq.update(leaves)
    for i in range(0, B, chunk):
        f = vit_ref.timm_vit_forward(images[i:i + chunk], Pq, heads=12, mona=mona)
        (f * dfi[i:i + chunk]).sum().backward()
    return fi, ft, float(loss), {k: v.grad for k, v in leaves.items()}, time.perf_counter() - t0


def logits_of(fi, ft, tau=0.07):
    """contrastive logits (reference losses.py:23-47): cosine similarities / temperature"""
    fi, ft = fi.detach().float().cpu(), ft.detach().float().cpu()
    fi = fi / fi.norm(dim=1, keepdim=True)
    ft = ft / ft.norm(dim=1, keepdim=True)
    return fi @ ft.t() / tau


def rms_rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.abs().max() + 1e-12))


FOLD = [True]


def run_case(B, variant, stress, chunk, threads, adapters="scaled"):
    from uia_hip import functional as UF
    from src.adapters import inject_mona_variant_to_open_clip
    from src.losses import InfoNCELoss
    from src.third_party.biomedclip.model import create_biomedclip
    UF.set_compute_dtype(torch.bfloat16)
    UF.set_ln_fold(FOLD[0])
    UF.reset_ln_flag()
    g = torch.Generator().manual_seed(41 + B)
    model = create_biomedclip(seed=3)
    for p in model.parameters():
        p.requires_grad_(False)
    inject_mona_variant_to_open_clip(model, variant=variant, bottleneck_dim=64)
    if adapters == "scaled":
        scale_adapters(model, g)
    if stress:
        stress_weights(model)
    for k, p in model.named_parameters():
        p.requires_grad_("mona" in k)
    model.eval()                                                          # dropout off for parity
    images, ids = torch.rand(B, 3, 224, 224, generator=g), captions(g, B)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainable = [k for k in P if "mona" in k]
    fref, tref, lref, gref, cpu_s = oracle_step(P, trainable, images, ids, variant, chunk, threads)

    dev = torch.device("cuda", 0)
    model = model.to(dev)
    from uia_hip import engine
    g3 = engine.GRAD_RESID3 and engine._hook_free(model)                  # what engine.contrastive_step (and bench.py) run with: three-byte residual gradients between the backward Functions
    UF.set_grad_resid3(g3)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        fi = model.encode_image(images.to(dev))
        ft = model.encode_text(ids.to(dev))
        loss = InfoNCELoss(0.07)(fi, ft)
        loss.backward()
        flag = UF.poll_ln_flag(sync=True)
    fold_tripped = bool(flag & 1)
    params = dict(model.named_parameters())
    per = {k: rel(params[k].grad, gref[k]) for k in trainable}
    worst = max(per, key=per.get)
    # per tensor against the GLOBAL gradient scale: a tensor whose own gradient is 1e-4 of the largest one is bf16 rounding noise of the sums it is a difference of —
    # its error relative to ITSELF says nothing (noise_estimator biases: 68 % / 266 % in round 4's tables), its error relative to the step's largest gradient does
    gmax = max(float(gref[k].abs().max()) for k in trainable)
    scaled = {k: float((params[k].grad.detach().float().cpu() - gref[k]).abs().max()) / gmax for k in trainable}
    own = {k: float(gref[k].abs().max()) / gmax for k in trainable}
    worst_scaled = max(scaled, key=scaled.get)
    got = torch.cat([params[k].grad.detach().float().cpu().flatten() for k in trainable])
    want = torch.cat([gref[k].flatten() for k in trainable])
    lg, lgr = logits_of(fi, ft), logits_of(fref, tref)
    out = {"B": B, "variant": variant, "stress": stress, "adapters": adapters, "image_features_rel": rel(fi, fref), "text_features_rel": rel(ft, tref),
           "image_features_rms_rel": rms_rel(fi, fref), "text_features_rms_rel": rms_rel(ft, tref),
           "logits_rel": float((lg - lgr).abs().max() / lgr.abs().max()), "logits_abs_max_err": float((lg - lgr).abs().max()), "logits_abs_max": float(lgr.abs().max()),
           "logits_spread_ref": float(lgr.max() - lgr.min()), "loss": float(loss), "loss_ref": lref,
           "grad_cosine": float(torch.dot(got, want) / (got.norm() * want.norm())), "grad_rel_l2": float((got - want).norm() / want.norm()),
           "grad_median_per_tensor_rel": sorted(per.values())[len(per) // 2], "grad_worst_per_tensor_rel": per[worst], "grad_worst_tensor": worst,
           "grad_worst_own_max_over_global_max": own[worst], "grad_worst_per_tensor_err_over_global_max": scaled[worst_scaled], "grad_worst_scaled_tensor": worst_scaled,
           "grad_worst_scaled_own_max_over_global_max": own[worst_scaled],
           "grad_per_tensor_table_top": sorted(((k, round(per[k], 4), round(own[k], 6), round(scaled[k], 6)) for k in trainable), key=lambda r: -r[3])[:8],
           "grad_resid3": bool(g3), "ln_fold_guard_tripped": fold_tripped, "warnings": [str(w.message)[:120] for w in wlist if "uia_hip" in str(w.message)],
           "oracle_cpu_seconds": round(cpu_s, 1), "oracle_threads": threads}
    if fold_tripped:                                                      # what the guard buys: the same step on the stand-alone LayerNorm kernels
        UF.clear_t_copies()
        for p in model.parameters():
            if p.grad is not None:
                p.grad = None
        assert not UF.ln_fold_enabled(torch.bfloat16)
        fi2, ft2 = model.encode_image(images.to(dev)), model.encode_text(ids.to(dev))
        out["after_guard_image_features_rel"], out["after_guard_text_features_rel"] = rel(fi2, fref), rel(ft2, tref)
    UF.set_ln_fold(True)
    UF.reset_ln_flag()
    UF.clear_t_copies()
    del model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="64,256")
    ap.add_argument("--variant", default="freq_enhanced")
    ap.add_argument("--stress", action="store_true")
    ap.add_argument("--no-ln-fold", action="store_true", help="the stand-alone LayerNorm kernels instead of the fold (A/B of the fold's share of the error)")
    ap.add_argument("--adapters", default="scaled", choices=["scaled", "init"], help="scaled: adapters away from their init so that every gradient path carries "
                    "signal (the parity tests' setting); init: as injected (what a fine-tune run starts from and what bench.py times)")
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_bench_batch.json"))
    args = ap.parse_args()
    FOLD[0] = not args.no_ln_fold
    threads = max(1, min(32, os.cpu_count() or 1))
    res = {}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    for B in [int(b) for b in args.batches.split(",") if b]:
        r = run_case(B, args.variant, False, args.chunk, threads, args.adapters)
        res[f"biomedclip_vitb16_mona_{args.variant}_bf16_B{B}" + ("" if args.adapters == "scaled" else "_adapters_at_init")] = r
        print(json.dumps(r), flush=True)
        json.dump(res, open(args.out, "w"), indent=1, sort_keys=True)
    if args.stress:
        r = run_case(64, args.variant, True, args.chunk, threads)
        res[f"biomedclip_vitb16_mona_{args.variant}_bf16_B64_outlier_stress"] = r
        print(json.dumps(r), flush=True)
        json.dump(res, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
