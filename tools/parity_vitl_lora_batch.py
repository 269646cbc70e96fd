"""BASELINE configs[4]'s geometry at a batch that reaches the large-M kernels: in-tree CLIP ViT-L/14 image tower (24 blocks, width 1024, 16 heads, 257 tokens) with
inject_lora_to_clip(r = 16, alpha = 32) on q, k, v, o of every block, bf16 mode, B = 32 (8 224 token rows: the 256-row ring tiles with their split-K M tail, the N = 64 and
K = 64 stream kernels, the one-node attention half) against oracle/vit_ref.py on the host cores: features, the whole LoRA gradient (cosine, relative L2, median and worst
per-tensor error).  Dropout off (the oracle takes no mask at model level; the kernels' masks are pinned by tests/test_round2_gpu.py and tests/test_round3_gpu.py).
`run_case()` is what tests/test_round6_gpu.py calls inside `pytest -m gpu` (round 6: full-batch parity of the secondary configurations is driver-visible).

    python tools/parity_vitl_lora_batch.py [--batch 32] [--out gpurun_out/parity_vitl_lora_batch.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from oracle import vit_ref


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def run_case(batch=32, chunk=8, threads=None):
    from uia_hip import functional as UF
    from src.adapters import inject_lora_to_clip
    from src.third_party.openai_clip.model import CLIP
    torch.set_num_threads(threads or max(1, min(32, os.cpu_count() or 1)))
    UF.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(59)
    torch.manual_seed(59)
    model = CLIP(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 1).eval()
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif k.endswith("weight") and "ln" in k:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    for p in model.parameters():
        p.requires_grad_(False)
    model, n = inject_lora_to_clip(model, lora_r=16, lora_alpha=32, lora_dropout=0.0)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "lora" in k:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    for k, p in model.named_parameters():
        p.requires_grad_("lora" in k)
    model.eval()
    B = batch
    images = torch.rand(B, 3, 224, 224, generator=g)
    dfeat = torch.randn(B, 768, generator=g)                                  # a fixed cotangent: the loss is <features, dfeat>, summed over chunks exactly
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    trainable = [k for k, p in model.named_parameters() if p.requires_grad]
    leaves = {k: P[k].clone().requires_grad_(True) for k in trainable}
    Pq = dict(P)
    Pq.update(leaves)
    t0 = time.perf_counter()
    feats = []
    for i in range(0, B, chunk):
        f = vit_ref.openai_vit_forward(images[i:i + chunk], Pq, heads=16, lora=dict(r=16, alpha=32))
        (f * dfeat[i:i + chunk]).sum().backward()
        feats.append(f.detach())
    fr = torch.cat(feats)
    cpu_s = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    model = model.to(dev)
    fi = model.encode_image(images.to(dev))
    (fi * dfeat.to(dev)).sum().backward()
    params = dict(model.named_parameters())
    gmax = max(float(v.grad.abs().max()) for v in leaves.values())
    per = {k: rel(params[k].grad, leaves[k].grad) for k in trainable if float(leaves[k].grad.abs().max()) > 1e-3 * gmax}
    worst = max(per, key=per.get)
    got = torch.cat([params[k].grad.detach().float().cpu().flatten() for k in trainable])
    want = torch.cat([leaves[k].grad.flatten() for k in trainable])
    res = {"B": B, "token_rows": B * 257, "lora_layers": n, "features_rel": rel(fi, fr),
           "features_rms_rel": float((fi.detach().float().cpu() - fr).pow(2).mean().sqrt() / fr.abs().max()),
           "grad_cosine": float(torch.dot(got, want) / (got.norm() * want.norm())), "grad_rel_l2": float((got - want).norm() / want.norm()),
           "grad_median_per_tensor_rel": sorted(per.values())[len(per) // 2], "grad_worst_per_tensor_rel": per[worst], "grad_worst_tensor": worst,
           "tensors_compared": len(per), "oracle_cpu_seconds": round(cpu_s, 1),
           "worst_six": [{"tensor": k, "rel": round(per[k], 4), "own_max_over_global_max": round(float(leaves[k].grad.abs().max()) / gmax, 5)}
                         for k in sorted(per, key=per.get, reverse=True)[:6]]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_vitl_lora_batch.json"))
    args = ap.parse_args()
    res = run_case(args.batch, args.chunk)
    B = args.batch
    print(json.dumps(res), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({f"clip_vit_l14_lora_r16_bf16_B{B}": res}, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
