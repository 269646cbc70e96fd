"""BiomedCLIP zero-shot classification on the MI355X HIP path — counterpart of
/root/reference/src/models/biomedclip/zero_shot.py (BASELINE configs[0]: the reference's own CPU-runnable case).

Kept from the reference: the command line (:31-72), adapter loading by parameter NAME from a fine-tune checkpoint
(`mona_state_dict` / `lora_state_dict` wrapper or a bare dict, :107-147 — names are the checkpoint wire format), eval mode,
and the scoring rule (:176-222): class prototypes = L2-normalised text features of a prompt ensemble per class, logits[b, c] =
mean over the class's prompts of 100 · <image feature, prompt feature>.  Host-side reporting (ROC figure, CSV, MONAI/
torchmetrics accumulators, :224-290) is outside the hot path; this module returns accuracy, AUC and cross-entropy computed
with a few torch ops.  The prompt ensembles are the reference's (src/models/zero_shot_prompt.py:29-54, carried as data: ten prompts per class, picked by
--dataset as in reference :168-173); --prompts_json overrides them.  Data: --synthetic or --data_pt {"images","labels"}.
"""
import argparse
import json
import logging
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import torch

from src.adapters import inject_lora_to_biomedclip, inject_mona_variant_to_open_clip
from src.models.zero_shot_prompt import ensemble_for
from src.third_party.biomedclip.model import SyntheticTokenizer, create_biomedclip
from src.utils.tools import parse_config, setup_logging
from uia_hip import functional as UF

LESION_TYPES = ("benign", "malignant")


def get_args(argv=None):
    p = argparse.ArgumentParser("Adaptation of Visual Foundation Model for Medical Ultrasound Image Analysis")
    p.add_argument("--exp", type=str, default="biomedclip_zero_shot")
    p.add_argument("--dataset", type=str, default="LN-INT")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--patch_size", type=int, default=16)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--strong_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--weak_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--mona_weights", type=str, default=None)
    p.add_argument("--mona_variant", type=str, default="freq_enhanced", choices=["baseline", "fractional", "noise_aware", "freq_enhanced", "hybrid"])
    p.add_argument("--mona_bottleneck", type=int, default=64)
    p.add_argument("--mona_layers", type=int, default=None)
    p.add_argument("--lora_weights", type=str, default=None)
    p.add_argument("--lora_r", type=int, default=16)
    p.add_argument("--lora_alpha", type=int, default=32)
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--num_classes", type=int, default=2)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--device", type=str, default="cuda:0" if torch.cuda.is_available() else "cpu")
    # additions of this build
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_test", type=int, default=64)
    p.add_argument("--data_pt", type=str, default=None, help=".pt with {'images': [N,3,S,S], 'labels': [N]}")
    p.add_argument("--prompts_json", type=str, default=None, help='{"benign": [...], "malignant": [...]}')
    p.add_argument("--ckpt_path", type=str, default=None)
    p.add_argument("--model_config", type=str, default=None)
    return p.parse_args(argv)


def load_adapter_by_name(model, path, wrapper_key):
    """reference :107-119 / :136-147: copy every checkpoint tensor whose name exists in the model; at least one must."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    state = ckpt.get(wrapper_key, ckpt)
    model_dict = model.state_dict()
    loaded = 0
    for name, param in state.items():
        if name in model_dict:
            model_dict[name] = param
            loaded += 1
    assert loaded > 0, f"No adapter parameters loaded from {path}"
    model.load_state_dict(model_dict)
    UF.WEIGHTS.bump()                                   # operand copies of the replaced weights are stale
    return loaded


def prepare_model(args):
    cfg = parse_config(args.model_config) if args.model_config else None
    state = torch.load(args.ckpt_path, map_location="cpu") if args.ckpt_path else None
    model = create_biomedclip(state_dict=state, config=cfg, seed=args.seed)
    tokenizer = SyntheticTokenizer(256 if cfg is None else cfg["text_cfg"]["max_position_embeddings"])
    if args.lora_weights:
        inject_lora_to_biomedclip(model, lora_r=args.lora_r, lora_alpha=args.lora_alpha, lora_dropout=0.0)
        n = load_adapter_by_name(model, args.lora_weights, "lora_state_dict")
        logging.info(f"✓ Loaded {n} LoRA parameters from {args.lora_weights}")
    elif args.mona_weights:
        inject_mona_variant_to_open_clip(model, variant=args.mona_variant, bottleneck_dim=args.mona_bottleneck, num_layers=args.mona_layers)
        n = load_adapter_by_name(model, args.mona_weights, "mona_state_dict")
        logging.info(f"✓ Loaded {n} MONA parameters from {args.mona_weights}")
    for p in model.parameters():
        p.requires_grad = False
    model.float()
    model.to(args.device)
    model.eval()
    return model, tokenizer


def _test_batches(args):
    if args.data_pt:
        blob = torch.load(args.data_pt)
        images, labels = blob["images"], blob["labels"].long()
        images = images.float() / 255.0 if images.dtype == torch.uint8 else images.float()
    elif args.synthetic:
        g = torch.Generator().manual_seed(args.seed)
        images = torch.rand(args.synthetic_test, 1, args.img_size, args.img_size, generator=g).repeat(1, 3, 1, 1)
        labels = torch.randint(0, 2, (args.synthetic_test,), generator=g)
    else:
        raise RuntimeError("no dataset: pass --synthetic or --data_pt")
    for i in range(0, len(labels), args.batch_size):
        yield images[i:i + args.batch_size], labels[i:i + args.batch_size]


def binary_auc(score, label):
    """Area under the ROC curve by the rank statistic (ties get the average rank)."""
    pos, neg = score[label == 1], score[label == 0]
    if len(pos) == 0 or len(neg) == 0:
        return float("nan")
    cmp = (pos[:, None] > neg[None, :]).double() + 0.5 * (pos[:, None] == neg[None, :]).double()
    return float(cmp.mean())


@torch.no_grad()
def class_text_features(model, tokenizer, prompts, device):
    feats = {}
    for c in LESION_TYPES:
        f = model.encode_text(tokenizer(prompts[c]).to(device))
        feats[c] = f / f.norm(dim=-1, keepdim=True)
    return feats


@torch.no_grad()
def ensemble_logits(model, images, text_feats):
    f = model.encode_image(images)
    f = f / f.norm(dim=-1, keepdim=True)
    return torch.stack([(100.0 * f @ text_feats[c].T).mean(dim=1) for c in LESION_TYPES], dim=1)      # [B, 2]


@torch.no_grad()
def test(args):
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model, tokenizer = prepare_model(args)
    prompts = json.load(open(args.prompts_json)) if args.prompts_json else ensemble_for(args.dataset)
    text_feats = class_text_features(model, tokenizer, prompts, args.device)
    proto_sim = float(text_feats["benign"].mean(0) @ text_feats["malignant"].mean(0))
    if proto_sim > 0.95:
        logging.warning(f"Text prompts very similar: {proto_sim:.4f}")
    all_logits, all_labels = [], []
    for images, labels in _test_batches(args):
        all_logits.append(ensemble_logits(model, images.to(args.device), text_feats).float().cpu())
        all_labels.append(labels)
    logits, labels = torch.cat(all_logits), torch.cat(all_labels)
    stats = {"acc": float((logits.argmax(1) == labels).float().mean()),
             "auc": binary_auc(torch.softmax(logits, dim=1)[:, 1], labels),
             "loss": float(torch.nn.functional.cross_entropy(logits, labels)),
             "n": int(len(labels))}
    logging.info(f"zero-shot {args.dataset}: acc {stats['acc'] * 100:.2f}  auc {stats['auc']:.4f}  loss {stats['loss']:.4f}  (n={stats['n']})")
    return stats, logits


def main(argv=None):
    args = get_args(argv)
    torch.manual_seed(args.seed)
    args.test_snapshot_path = f"runs/{args.exp}/{args.dataset}/test"
    os.makedirs(args.test_snapshot_path, exist_ok=True)
    setup_logging(args, args.test_snapshot_path)
    stats, _ = test(args)
    with open(os.path.join(args.test_snapshot_path, "results.json"), "w") as f:
        json.dump(stats, f)
    return stats


if __name__ == "__main__":
    main()
