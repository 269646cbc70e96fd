"""UniMed-CLIP contrastive fine-tuning with Mona adapters on the MI355X HIP path — drop-in for /root/reference/src/models/unimedclip/finetune.py.

Same command line (every flag and default of reference :27-63: --version ViT-B-16-quickgelu, --ckpt ckpt/unimed_clip_vit_b16.pt, default --mona_variant noise_aware, batch 64,
1000 epochs) and the reference's loop (:111-305), which is the MetaCLIP entry point's loop verbatim — so it runs on src.models.metaclip.finetune.train (engine.ContrastiveLoop: the
measured step) with this family's model preparation (:66-108):
  * open_clip's NATIVE ViT-B/16 (QuickGELU forced, :68-73) — src/third_party/open_clip/model.NativeCLIP: batch-first blocks under visual.transformer.resblocks, image_size /
    patch_size / grid_size on the tower, the case-2 branch of inject_mona_variant_to_open_clip (mona.py:633-676);
  * the checkpoint's `state_dict` (or the dict itself), `module.` prefixes stripped (:77-81), and ONLY the `visual.*` keys and `logit_scale` loaded, strict=False (:83-86) —
    the text tower keeps its random initialisation, exactly as in the reference (a quirk recorded in SURVEY Appendix C);
  * every parameter frozen, the adapters injected, parameters whose lower-cased name contains "mona" trainable, fp32 masters (:89-106).
Differences forced by the build image: open_clip's HFTokenizer("microsoft/BiomedNLP-BiomedBERT-base-uncased-abstract", context_length=77) needs the network — captions are
tokenised by the deterministic BERT-style stand-in (CLS 2, SEP 3, ids 1000-30000, pad 0; context 77); without a checkpoint file the tower is randomly initialised.
Added, non-breaking: --dtype, --synthetic / --data_pt, --model_config, data parallelism under torch.distributed.run."""
import argparse
import logging
import os
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import numpy as np
import torch

from src.adapters import inject_mona_variant_to_open_clip
from src.models.metaclip import finetune as _loop
from src.third_party.biomedclip.model import SyntheticTokenizer
from src.third_party.open_clip.model import create_native_clip
from src.utils.tools import default_device, parse_config, setup_logging


def get_args(argv=None):
    p = argparse.ArgumentParser("UniMedCLIP Fine-tuning with Frequency-Enhanced MONA")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--strong_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--weak_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--version", type=str, default="ViT-B-16-quickgelu")
    p.add_argument("--ckpt", type=str, default="ckpt/unimed_clip_vit_b16.pt")
    p.add_argument("--mona_variant", type=str, default="noise_aware")
    p.add_argument("--exp", type=str, default="unimedclip_finetune")
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
    p.add_argument("--model_config", type=str, default=None, help="python dict literal: keyword arguments of src.third_party.open_clip.model.NativeCLIP (tests)")
    return p.parse_args(argv)


def _config(args):
    return parse_config(args.model_config) if args.model_config else None


def make_tokenizer(args):
    cfg = _config(args) or {}
    return SyntheticTokenizer(cfg.get("context_length", 77))


def prepare_model(args):
    """reference :66-108."""
    model = create_native_clip(args.version, config=_config(args), seed=args.seed, force_quick_gelu=True)
    if args.ckpt and os.path.exists(args.ckpt):
        checkpoint = torch.load(args.ckpt, map_location="cpu", weights_only=False)
        state_dict = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
        state_dict = {k.replace("module.", ""): v for k, v in state_dict.items()}
        visual_state_dict = {k: v for k, v in state_dict.items() if k.startswith("visual.") or k == "logit_scale"}
        model.load_state_dict(visual_state_dict, strict=False)
        logging.info(f"loaded {len(visual_state_dict)} visual tensors from {args.ckpt}")
    else:
        logging.info(f"checkpoint {args.ckpt} not found: randomly initialised {args.version}")
    for param in model.parameters():
        param.requires_grad = False
    model, mona_count = inject_mona_variant_to_open_clip(model, variant=args.mona_variant, bottleneck_dim=args.mona_bottleneck, num_layers=args.mona_layers)
    for name, param in model.named_parameters():
        if "mona" in name.lower():
            param.requires_grad = True
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
