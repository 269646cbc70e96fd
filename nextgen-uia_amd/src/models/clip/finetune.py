"""OpenAI-CLIP contrastive fine-tuning with Mona adapters on the MI355X HIP path — drop-in for /root/reference/src/models/clip/finetune.py.

Same command line (every flag and default of reference :27-62: default --mona_variant noise_aware, --ckpt ckpt/ViT-B-16.pt, batch 64, 1000 epochs, no accumulation) and the
reference's loop (:92-215), which is the MetaCLIP entry point's loop verbatim — per batch InfoNCE on L2-normalised features, non-finite batches skipped with their update,
clip_grad_norm_(1.0) -> AdamW -> cosine LR every iteration, eval-mode validation per epoch, best-val checkpoint of the "mona" parameters, early stopping — so it runs on
src.models.metaclip.finetune.train (engine.ContrastiveLoop: the measured step, no host read per batch) with this family's model preparation (:65-89): the OpenAI-layout CLIP of
src/third_party/openai_clip/model.py (sequence-first blocks, nn.MultiheadAttention names, QuickGELU, eps 1e-5) + inject_mona_variant_to_clip.

Differences forced by the build image: `clip.load` (torchvision transforms, the BPE vocabulary file) is not available — --ckpt is read as an OpenAI TorchScript archive or a
plain state dict when the file exists, otherwise the ViT-B/16 geometry is randomly initialised; captions are tokenised by the deterministic stand-in tokenizer (context 77,
<start>/<end> ids of the CLIP vocabulary).  Added, non-breaking: --dtype, --synthetic / --data_pt, --model_config, data parallelism under torch.distributed.run."""
import argparse
import logging
import os
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import numpy as np
import torch

from src.adapters import inject_mona_variant_to_clip
from src.models.metaclip import finetune as _loop
from src.third_party.open_clip.model import SyntheticClipTokenizer
from src.third_party.openai_clip.model import CLIP, build_model
from src.utils.tools import default_device, parse_config, setup_logging


def get_args(argv=None):
    p = argparse.ArgumentParser("CLIP Fine-tuning with MONA adapters and MoHN Loss")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--strong_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--weak_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--mona_variant", type=str, default="noise_aware")
    p.add_argument("--exp", type=str, default="clip_finetune")
    p.add_argument("--ckpt", type=str, default="ckpt/ViT-B-16.pt")
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--mona_bottleneck", type=int, default=64)
    p.add_argument("--mona_layers", type=int, default=None)
    p.add_argument("--temperature", type=float, default=0.07)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--epochs", type=int, default=1000)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--lr_min", type=float, default=1e-8)
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--beta1_adam", type=float, default=0.9)
    p.add_argument("--beta2_adam", type=float, default=0.95)
    p.add_argument("--device", type=str, default=default_device())       # decided without a HIP call: the loader workers fork first
    p.add_argument("--patience", type=int, default=10)
    # additions of this build
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_train", type=int, default=512)
    p.add_argument("--synthetic_val", type=int, default=128)
    p.add_argument("--data_pt", type=str, default=None)
    p.add_argument("--model_config", type=str, default=None, help="python literal: positional arguments of src.third_party.openai_clip.model.CLIP (tests)")
    return p.parse_args(argv)


def _geometry(args):
    return tuple(parse_config(args.model_config)) if args.model_config else (512, 224, 12, 768, 16, 77, 49408, 512, 8, 12)     # ViT-B/16 (reference --ckpt default)


def make_tokenizer(args):
    geo = _geometry(args)
    return SyntheticClipTokenizer(geo[5], geo[6])


def prepare_model(args):
    """reference :65-89: load CLIP, freeze everything, inject the Mona adapters, train only "mona" parameters, float32 masters."""
    if args.ckpt and os.path.exists(args.ckpt):
        try:
            sd = torch.jit.load(args.ckpt, map_location="cpu").state_dict()        # OpenAI's released checkpoints are TorchScript archives (clip.py:128-136 of the reference)
        except RuntimeError:
            sd = torch.load(args.ckpt, map_location="cpu")
        model = build_model(sd)
    else:
        logging.info(f"checkpoint {args.ckpt} not found: randomly initialised CLIP {_geometry(args)}")
        torch.manual_seed(args.seed)
        model = CLIP(*_geometry(args))
    for p in model.parameters():
        p.requires_grad = False
    model, mona_count = inject_mona_variant_to_clip(model, variant=args.mona_variant, bottleneck_dim=args.mona_bottleneck, num_layers=args.mona_layers)
    for name, p in model.named_parameters():
        if "mona" in name.lower():
            p.requires_grad = True
    model.float()
    model.to(args.device)
    return model, make_tokenizer(args)


def train(args):
    return _loop.train(args, prepare=prepare_model, tokenizer_of=make_tokenizer)


def main(argv=None):
    args = get_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.train_snapshot_path = f"runs/{args.exp}"
    os.makedirs(args.train_snapshot_path, exist_ok=True)
    setup_logging(args, args.train_snapshot_path)
    return train(args)


if __name__ == "__main__":
    main()
