"""BiomedCLIP segmentation with the feature-pyramid adapter on the HIP path — counterpart of
/root/reference/src/models/biomedclip/segmentation.py.

Kept: the command line (:28-72; default --mona_variant hybrid, batch 32, AdamW betas 0.9/0.95, no gradient clipping), model
assembly (:81-135: optional LoRA or Mona adapters loaded BY NAME from a fine-tune checkpoint, TimmCLIPAdapter(task="seg") on
layers 3/6/9, freeze_clip_backbone()), the loop (:138-255): DiceCE(to_onehot_y, softmax, squared_pred) per iteration with a
per-iteration cosine schedule, validation by mean foreground Dice every 10 epochs and at the last one, early stopping by
--patience, and the checkpoint dict {"reduces", "blocks", "seg_head", "mona"} (:212-225) under runs/<exp>/<dataset>/train.
Left out: TensorBoard images, MONAI HD95/ASD and the post-training test pass (:257-330) — host-side reporting.
Data: `--synthetic` (grayscale-repeated U[0,1) images with random-ellipse masks) or `--data_pt {"images","labels"}`.
"""
import argparse
import logging
import os
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import numpy as np
import torch

from src.adapters import inject_lora_to_biomedclip, inject_mona_variant_to_open_clip
from src.losses.dice import DiceCELoss, dice_per_image
from src.models.biomedclip.zero_shot import load_adapter_by_name
from src.models.clipseg.segmentation import synthetic_batch
from src.third_party.biomedclip.model import create_biomedclip
from src.third_party.timm.clip_adapter import TimmCLIPAdapter
from src.utils.tools import model_summary, parse_config, setup_logging
from uia_hip import functional as UF
from uia_hip.engine import FlatAdapterOptimizer, bind_device, cosine_lr, init_data_parallel


def get_args(argv=None):
    p = argparse.ArgumentParser("Adaptation of Visual Foundation Model for Medical Ultrasound Image Analysis")
    p.add_argument("--exp", type=str, default="biomedclip_seg")
    p.add_argument("--dataset", type=str, default="LN-INT")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--patch_size", type=int, default=16)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--strong_augs", default=True, action=argparse.BooleanOptionalAction)
    p.add_argument("--weak_augs", default=True, action=argparse.BooleanOptionalAction)
    p.add_argument("--mona_variant", type=str, default="hybrid")
    p.add_argument("--mona_weights", type=str, default=None)
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--reduce_dim", type=int, default=512)
    p.add_argument("--mona_bottleneck", type=int, default=64)
    p.add_argument("--mona_layers", type=int, default=None)
    p.add_argument("--lora_weights", type=str, default=None)
    p.add_argument("--lora_r", type=int, default=16)
    p.add_argument("--lora_alpha", type=int, default=32)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--lr_min", type=float, default=1e-8)
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.95)
    p.add_argument("--device", type=str, default="cuda:0" if torch.cuda.is_available() else "cpu")
    p.add_argument("--patience", type=int, default=15)
    p.add_argument("--test", default=False, action="store_true")
    # additions of this build
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_train", type=int, default=256)
    p.add_argument("--synthetic_val", type=int, default=64)
    p.add_argument("--data_pt", type=str, default=None)
    p.add_argument("--ckpt_path", type=str, default=None, help="open_clip BiomedCLIP state dict (.pt); random init if absent")
    p.add_argument("--model_config", type=str, default=None)
    p.add_argument("--extract_layers", type=str, default="3,6,9", help="transformer blocks tapped by the adapter (reference: fixed 3,6,9)")
    p.add_argument("--val_every", type=int, default=10, help="epochs between validations (reference: fixed 10)")
    return p.parse_args(argv)


def prepare_model(args):
    cfg = parse_config(args.model_config) if args.model_config else None
    state = torch.load(args.ckpt_path, map_location="cpu") if args.ckpt_path else None
    clip_model = create_biomedclip(state_dict=state, config=cfg, seed=args.seed)
    clip_model.float()
    if args.lora_weights:
        inject_lora_to_biomedclip(clip_model, lora_r=args.lora_r, lora_alpha=args.lora_alpha, lora_dropout=0.0)
        n = load_adapter_by_name(clip_model, args.lora_weights, "lora_state_dict")
        logging.info(f"✓ Loaded {n} pretrained LoRA parameters from {args.lora_weights}")
    elif args.mona_weights:
        inject_mona_variant_to_open_clip(clip_model, variant=args.mona_variant, bottleneck_dim=args.mona_bottleneck, num_layers=args.mona_layers)
        n = load_adapter_by_name(clip_model, args.mona_weights, "mona_state_dict")
        logging.info(f"✓ Loaded {n} pretrained MONA parameters from {args.mona_weights}")
    adapter = TimmCLIPAdapter(clip_model=clip_model, extract_layers=[int(v) for v in args.extract_layers.split(",")], reduce_dim=args.reduce_dim,
                              num_classes=args.num_classes, img_size=args.img_size, patch_size=args.patch_size, task="seg")
    adapter.to(args.device)
    adapter.freeze_clip_backbone()
    return adapter


def checkpoint_dict(model):
    """reference :212-225: adapter heads as module state dicts plus the backbone's Mona parameters by full name."""
    return {"reduces": model.reduces.state_dict(), "blocks": model.blocks.state_dict(), "seg_head": model.seg_head.state_dict(),
            "mona": {n: p.data.clone() for n, p in model.named_parameters() if "mona" in n}}


def _batches(args, n, seed0, rank=0, world=1):
    if args.data_pt:
        blob = torch.load(args.data_pt)
        images, labels = blob["images"].float(), blob["labels"].float()
        nb = len(images) // args.batch_size // world * world                  # every rank takes the same number of batches, its own ones
        for b in range(rank, nb, world):
            i = b * args.batch_size
            yield images[i:i + args.batch_size].to(args.device), labels[i:i + args.batch_size].to(args.device)
        return
    if not args.synthetic:
        raise RuntimeError("no dataset: pass --synthetic or --data_pt (the reference's PIL/torchvision loaders are outside this build)")
    for i in range(max(1, n // args.batch_size)):
        yield synthetic_batch(args.batch_size, args.img_size, seed0 + i * world + rank, args.device)


def train(args):
    rank, _, world = bind_device(args)                         # data parallel: cuda:LOCAL_RANK before anything is allocated
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    UF.set_dropout_seed(args.seed + 7919 * rank)
    model = prepare_model(args)
    model.train()
    logging.info(model_summary({"model": model}))
    criterion = DiceCELoss()
    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=args.lr, betas=(args.beta1, args.beta2),
                               weight_decay=args.weight_decay, max_norm=0.0)
    if world > 1:
        init_data_parallel(opt)
    iters_per_epoch = len(list(_batches(args, args.synthetic_train, 0, rank, world))) if args.data_pt else max(1, args.synthetic_train // args.batch_size)
    max_iters = iters_per_epoch * args.epochs
    iter_num, best_val_dice, patience, last = 0, 0.0, 0, None
    for epoch in range(args.epochs):
        for images, labels in _batches(args, args.synthetic_train, args.seed * 7919 + epoch * 100003, rank, world):
            opt.zero_grad()
            preds = model(images)
            loss = criterion(preds, labels)
            loss.backward()
            opt.all_reduce()
            opt.step(lr=cosine_lr(args.lr, args.lr_min, iter_num, max_iters))
            UF.clear_t_copies()
            iter_num += 1
            last = float(loss) if iter_num % 10 == 0 or last is None else last
        if (epoch > 0 and epoch % args.val_every == 0) or (epoch == args.epochs - 1):
            model.eval()
            dices, vloss = [], []
            with torch.no_grad():
                for images, labels in _batches(args, args.synthetic_val, args.seed * 104729 + 17):
                    preds = model(images)
                    vloss.append(float(criterion(preds, labels)))
                    dices.append(dice_per_image(preds, labels))
            dice_mean = float(torch.nanmean(torch.cat(dices)))
            logging.info(f"\titer: {iter_num}, loss: {np.mean(vloss):.4f}, dice: {dice_mean * 100:.2f}")
            if dice_mean > best_val_dice:
                patience, best_val_dice = 0, dice_mean
                if rank == 0:
                    torch.save(checkpoint_dict(model), os.path.join(args.train_snapshot_path, "best_model.pth"))
            else:
                patience += 1
            if patience >= args.patience:
                logging.info(f"\nEarly stopping at epoch {epoch + 1}")
                break
            model.train()
    return {"iters": iter_num, "best_val_dice": best_val_dice, "last_loss": last}


def main(argv=None):
    args = get_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.train_snapshot_path = f"runs/{args.exp}/{args.dataset}/train"
    os.makedirs(args.train_snapshot_path, exist_ok=True)
    setup_logging(args, args.train_snapshot_path)
    if args.test:
        raise NotImplementedError("--test (metrics over a held-out set with MONAI HD95/ASD) is host-side reporting outside this build")
    return train(args)


if __name__ == "__main__":
    main()
