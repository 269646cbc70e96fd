"""CLIPSeg segmentation training on the HIP path — counterpart of /root/reference/src/models/clipseg/segmentation.py.

Loop semantics kept (:106-236): frozen OpenAI CLIP ViT-B/16 + trainable CLIPSeg decoder, one fixed prompt per dataset repeated
over the batch (:142), DiceCE loss (:84), AdamW + per-iteration cosine schedule, `{"decoder": state_dict}` checkpoints
(:193-197), runs/<exp>/<dataset>/train layout.  CLI: the reference's flags (:28-66) plus --dtype, --synthetic, --clip_ckpt.
Data: the reference's PIL/torchvision segmentation datasets are host-side I/O outside the hot path; `--synthetic` supplies
grayscale-repeated U[0,1) images with random-ellipse masks."""
import argparse
import logging
import os
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import numpy as np
import torch

from src.losses.dice import DiceCELoss, dice_per_image
from src.models.clipseg.prompt import busi_prompt, synthetic_prompt
from src.third_party.openai_clip.clipseg_adapter import CLIPSegAdapter
from src.third_party.openai_clip.model import CLIP, build_model
from src.utils.tools import model_summary, setup_logging
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, bind_device, cosine_lr, init_data_parallel


def get_args(argv=None):
    p = argparse.ArgumentParser("CLIPSeg segmentation")
    p.add_argument("--dataset", type=str, default="BUSI")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--exp", type=str, default="clipseg")
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--reduce_dim", type=int, default=512, help="unused, as in the reference (the decoder's own 64 applies)")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--lr_min", type=float, default=1e-6)
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--device", type=str, default="cuda:0" if torch.cuda.is_available() else "cpu")
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_train", type=int, default=256)
    p.add_argument("--clip_ckpt", type=str, default=None, help="OpenAI ViT-B-16 state dict (.pt); random init if absent")
    p.add_argument("--decoder_ckpt", type=str, default=None, help="CIDAS/clipseg-rd64-refined decoder state dict")
    p.add_argument("--iters", type=int, default=None, help="stop after this many updates (benchmarking)")
    return p.parse_args(argv)


def synthetic_batch(B, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 1, size, size, generator=g).repeat(1, 3, 1, 1)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    c = torch.rand(B, 2, generator=g) * size * 0.5 + size * 0.25
    r = torch.rand(B, 2, generator=g) * size * 0.2 + size * 0.08
    mask = (((yy[None] - c[:, 0, None, None]) / r[:, 0, None, None]) ** 2 + ((xx[None] - c[:, 1, None, None]) / r[:, 1, None, None]) ** 2) <= 1
    return img.to(device), mask[:, None].float().to(device)


def prepare_model(args):
    if args.clip_ckpt:
        clip = build_model(torch.load(args.clip_ckpt, map_location="cpu"))
    else:
        torch.manual_seed(args.seed)
        clip = CLIP(512, args.img_size, 12, 768, 16, 77, 49408, 512, 8, 12)          # OpenAI ViT-B/16 geometry, random init
    clip.float()
    model = CLIPSegAdapter(clip)
    if args.decoder_ckpt:
        model.decoder.load_state_dict(torch.load(args.decoder_ckpt, map_location="cpu"))
    model.freeze_clip_backbone()
    return model.to(args.device)


def train(args):
    rank, _, world = bind_device(args)                         # data parallel: cuda:LOCAL_RANK before anything is allocated
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model = prepare_model(args)
    logging.info(model_summary({"model": model}))
    criterion = DiceCELoss()
    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=args.lr, betas=(0.9, 0.999),
                               weight_decay=args.weight_decay, max_norm=0.0)
    if world > 1:
        init_data_parallel(opt)
    prompt = (busi_prompt if args.dataset == "BUSI" else synthetic_prompt(seed=hash(args.dataset) % 1000)).to(args.device)
    if not args.synthetic:
        raise RuntimeError("no dataset: pass --synthetic (the reference's PIL/torchvision loaders are outside this build)")
    iters_per_epoch = max(1, args.synthetic_train // args.batch_size)
    total = args.iters or iters_per_epoch * args.epochs
    model.train()
    it, last = 0, None
    while it < total:
        images, labels = synthetic_batch(args.batch_size, args.img_size, args.seed * 7919 + it * world + rank, args.device)
        batch_prompt = prompt.repeat(images.shape[0], 1)                                  # segmentation.py:142
        opt.zero_grad()
        preds = model(images, input_ids=batch_prompt)
        loss = criterion(preds, labels)
        loss.backward()
        opt.all_reduce()
        opt.step(lr=cosine_lr(args.lr, args.lr_min, it, total))
        UF.clear_t_copies()
        it += 1
        if it % 10 == 0 or it == total:
            last = float(loss)
            logging.info(f"iter {it}/{total}: loss {last:.4f}, Dice {float(torch.nanmean(dice_per_image(preds.detach(), labels))):.4f}")
    if rank == 0:
        torch.save({"decoder": model.decoder.state_dict()}, os.path.join(args.snapshot_path, "best_model.pth"))
    return {"iters": it, "loss": last}


def main(argv=None):
    args = get_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.snapshot_path = f"runs/{args.exp}/{args.dataset}/train"
    os.makedirs(args.snapshot_path, exist_ok=True)
    setup_logging(args, args.snapshot_path)
    return train(args)


if __name__ == "__main__":
    main()
