"""BASELINE configs[3] at its own batch size: CLIPSeg (frozen OpenAI ViT-B/16 taps 3/6/9 + prompt text tower + FiLM decoder, 224 x 224) at
bs = 128, bf16 mode, against oracle/clipseg_ref.py on the host cores — "BUSI Dice parity" on synthetic ellipse masks (there is no dataset in the
build container): logits error, masks identical outside the logit-margin band, per-image Dice of the argmax masks (reference
src/utils/tools.py:185-206 semantics, empty ground truth -> NaN dropped), DiceCE loss, and the decoder gradient of one training step.
`run_case()` is what tests/test_round6_gpu.py calls inside `pytest -m gpu` (round 6: full-batch parity of the secondary configurations is driver-visible).

    python tools/parity_clipseg_batch.py [--batch 128] [--out gpurun_out/parity_clipseg_batch.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nextgen-uia_amd")]
import torch
from oracle import clipseg_ref, losses_ref


def run_case(batch=128, chunk=16, threads=None, dtype="bf16"):
    from uia_hip import functional as UF
    from src.losses.dice import DiceCELoss, dice_per_image
    from src.models.clipseg import segmentation as S
    from src.third_party.openai_clip.clipseg_adapter import CLIPSegAdapter, CLIPSegDecoder
    from src.third_party.openai_clip.model import CLIP
    torch.set_num_threads(threads or max(1, min(32, os.cpu_count() or 1)))
    UF.set_compute_dtype(torch.bfloat16 if dtype == "bf16" else torch.float32)
    g = torch.Generator().manual_seed(43)
    torch.manual_seed(43)
    clip = CLIP(512, 224, 12, 768, 16, 77, 49408, 512, 8, 12).eval()
    dec = CLIPSegDecoder(vision_hidden=768, projection_dim=512, reduce_dim=64, extract_layers=(3, 6, 9), heads=4, intermediate=2048, patch_size=16)
    model = CLIPSegAdapter(clip, decoder=dec)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.02 if k.startswith("clip_model.") else 0.05))
            elif "ln" in k.lower() or "norm" in k.lower():
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g)) if k.endswith("weight") else p.copy_(0.02 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    model.freeze_clip_backbone()
    B = batch
    images, labels = S.synthetic_batch(B, 224, 5, "cpu")
    prompt = S.busi_prompt.repeat(B, 1)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [k for k in P if k.startswith("decoder.")]
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    Pq = dict(P)
    Pq.update(leaves)
    t0 = time.perf_counter()
    refs = []
    # the DiceCE loss couples the images of a batch only through its mean: chunk losses weighted by their share give the batch loss and gradient
    lref = 0.0
    for i in range(0, B, chunk):
        r = clipseg_ref.adapter_forward(images[i:i + chunk], prompt[i:i + chunk], Pq, vit_heads=12, text_heads=8, extract_layers=(3, 6, 9))
        l = losses_ref.dice_ce(r, labels[i:i + chunk]) * (r.shape[0] / B)
        l.backward()
        lref += float(l)
        refs.append(r.detach())
    ref = torch.cat(refs)
    cpu_s = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    model = model.to(dev)
    out = model(images.to(dev), input_ids=prompt.to(dev))
    loss = DiceCELoss()(out, labels.to(dev))
    loss.backward()
    o = out.detach().float().cpu()
    e_out = float((o - ref).abs().max() / ref.abs().max())
    margin = (ref[:, 1] - ref[:, 0]).abs()
    disagree = o.argmax(1) != ref.argmax(1)
    thr = 2e-2 * float(ref.abs().max())
    d_gpu = dice_per_image(out.detach(), labels.to(dev)).float().cpu()
    d_ref = losses_ref.dice_metric(ref, labels)
    ok = ~torch.isnan(d_ref)
    params = dict(model.named_parameters())
    got = torch.cat([params[k].grad.detach().float().cpu().flatten() for k in names])
    want = torch.cat([leaves[k].grad.flatten() for k in names])
    res = {"B": B, "logits_rel": e_out, "dicece": float(loss), "dicece_ref": lref, "mask_pixels": int(disagree.numel()), "mask_pixels_disagreeing": int(disagree.sum()),
           "of_which_outside_margin_2pct": int((disagree & (margin >= thr)).sum()),
           "of_which_outside_abs_margin_1e-3": int((disagree & (margin >= 1e-3)).sum()), "dtype": dtype, "logits_absmax_ref": float(ref.abs().max()),
           "dice_mean": float(d_gpu[ok].mean()), "dice_mean_ref": float(d_ref[ok].mean()), "dice_max_abs_diff_per_image": float((d_gpu[ok] - d_ref[ok]).abs().max()),
           "images_with_ground_truth": int(ok.sum()),
           "grad_cosine": float(torch.dot(got, want) / (got.norm() * want.norm())), "grad_rel_l2": float((got - want).norm() / want.norm()),
           "oracle_cpu_seconds": round(cpu_s, 1)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_clipseg_batch.json"))
    args = ap.parse_args()
    res = run_case(args.batch, args.chunk, dtype=args.dtype)
    B = args.batch
    print(json.dumps(res), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({f"clipseg_vitb16_{args.dtype}_B{B}": res}, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
