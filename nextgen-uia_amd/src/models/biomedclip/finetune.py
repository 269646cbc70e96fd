"""BiomedCLIP contrastive fine-tuning on the MI355X HIP path — drop-in for /root/reference/src/models/biomedclip/finetune.py.

Same command line (every flag and default of reference :32-111), same loop semantics (:211-361): micro-batch InfoNCE,
(loss / accumulation_steps).backward(), non-finite batches skipped, clip_grad_norm_ → AdamW → cosine LR once per cycle,
validation each epoch, best-val checkpoint of the adapter parameters only (names containing "mona"/"lora", :200-208),
early stopping, runs/<exp>/{log.log,best_model.pth}.  Added, non-breaking: --dtype {bf16,fp32}, --synthetic / --data_pt,
--ckpt_path (an open_clip BiomedCLIP state dict; without it the towers are randomly initialised because the build image
has no network), and data parallelism when launched under torch.distributed.run (one process per GPU, one RCCL
all-reduce of the flat adapter-gradient buffer per optimiser update ≡ the reference's accumulation).

What differs underneath: encode_image / encode_text / loss / backward / optimiser run in libuia_hip.so (uia_hip.*).
`--method full` (the reference's default): the image tower trains through op-level functions with `uia_wgrad` weight gradients
(all blocks or --tune_layers last3/6/9, lr forced to 1e-6 as in reference :153-155, whole state dict checkpointed); with
--tune_text_encoder the BERT weights, LayerNorms and embedding tables train as well (`uia_embed_bwd`).
"""
import argparse
import logging
import math
import os
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import numpy as np
import torch

from src.adapters import inject_lora_to_biomedclip, inject_mona_variant_to_open_clip
from src.datasets import finetune as dataset_finetune
from src.losses import InfoNCELoss
from src.third_party.biomedclip.model import SyntheticTokenizer, create_biomedclip
from src.utils.tools import default_device, model_summary, parse_config, setup_logging
from uia_hip import functional as UF
from uia_hip.engine import ContrastiveLoop, DevicePrefetcher, FlatAdapterOptimizer, bind_device, dist_env, init_data_parallel, sum_over_ranks


def get_args(argv=None):
    p = argparse.ArgumentParser("BiomedCLIP Fine-tuning")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--strong_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--weak_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--exp", type=str, default="biomedclip_finetune")
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--ckpt", type=str, default=None, help="Path to finetuned model checkpoint")
    p.add_argument("--method", type=str, default="full", choices=["full", "mona", "lora"])
    p.add_argument("--tune_text_encoder", default=False, action="store_true")
    p.add_argument("--tune_layers", type=str, default="all", choices=["last3", "last6", "last9", "all"])
    p.add_argument("--mona_variant", type=str, default="freq_enhanced", choices=["baseline", "fractional", "noise_aware", "freq_enhanced", "hybrid"])
    p.add_argument("--mona_bottleneck", type=int, default=64)
    p.add_argument("--mona_layers", type=int, default=None)
    p.add_argument("--lora_r", type=int, default=16)
    p.add_argument("--lora_alpha", type=int, default=32)
    p.add_argument("--lora_dropout", type=float, default=0.1)
    p.add_argument("--lora_layers", type=int, default=None)
    p.add_argument("--temperature", type=float, default=0.07)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--epochs", type=int, default=32)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--lr_min", type=float, default=1e-8)
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--beta1_adam", type=float, default=0.9)
    p.add_argument("--beta2_adam", type=float, default=0.95)
    p.add_argument("--device", type=str, default=default_device())       # decided without a HIP call: the loader workers fork first
    p.add_argument("--patience", type=int, default=10)
    p.add_argument("--accumulation_steps", type=int, default=4)
    p.add_argument("--grad_clip", type=float, default=1.0)
    # additions of this build
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"], help="operand precision of the HIP kernels")
    p.add_argument("--synthetic", action="store_true", help="synthetic image-caption pairs (no dataset files needed)")
    p.add_argument("--synthetic_train", type=int, default=512)
    p.add_argument("--synthetic_val", type=int, default=128)
    p.add_argument("--data_pt", type=str, default=None, help=".pt with {'images': [N,3,S,S], 'texts': [str]}")
    p.add_argument("--ckpt_path", type=str, default=None, help="open_clip BiomedCLIP state dict (.pt); random init if absent")
    p.add_argument("--model_config", type=str, default=None, help="python dict literal overriding the BiomedCLIP geometry (tests)")
    p.add_argument("--stats_json", type=str, default=None, help="write train()'s return value (per-epoch wall time and update counts included) to this file")
    return p.parse_args(argv)


def make_tokenizer(args):
    cfg = parse_config(args.model_config) if args.model_config else None
    return SyntheticTokenizer(256 if cfg is None else cfg["text_cfg"]["max_position_embeddings"])


def prepare_model(args):
    cfg = parse_config(args.model_config) if args.model_config else None
    state = torch.load(args.ckpt_path, map_location="cpu") if args.ckpt_path else None
    model = create_biomedclip(state_dict=state, config=cfg, seed=args.seed)
    tokenizer = make_tokenizer(args)
    model.float()
    if args.method == "full":                                    # reference :134-157
        if not args.tune_text_encoder:
            for p in model.text.parameters():
                p.requires_grad = False
            logging.info("Text encoder frozen")
        if args.tune_layers != "all":
            blocks = model.visual.trunk.blocks
            for p in model.visual.parameters():
                p.requires_grad = False
            layers = {"last3": 3, "last6": 6, "last9": 9}.get(args.tune_layers, 0)
            for i in range(len(blocks) - layers, len(blocks)):
                for p in blocks[i].parameters():
                    p.requires_grad = True
            logging.info(f"Tuning ViT layers {len(blocks) - layers}-{len(blocks) - 1} ({layers}/{len(blocks)} layers)")
        if args.lr > 1e-5:
            args.lr = 1e-6
            logging.info(f"Adjusted learning rate to {args.lr} for full fine-tuning")
        if hasattr(model, "logit_scale"):
            model.logit_scale.requires_grad = False              # not used by the InfoNCE loss (fixed temperature): no gradient ever reaches it
        model.to(args.device)
        tr = sum(p.numel() for p in model.parameters() if p.requires_grad)
        tot = sum(p.numel() for p in model.parameters())
        logging.info(f"Trainable parameters: {tr:,} / {tot:,} ({100 * tr / tot:.2f}%)")
        return model, tokenizer
    for p in model.parameters():
        p.requires_grad = False
    if args.method == "mona":
        inject_mona_variant_to_open_clip(model, variant=args.mona_variant, bottleneck_dim=args.mona_bottleneck, num_layers=args.mona_layers)
        key = "mona"
        logging.info(f"MONA variant: {args.mona_variant}, bottleneck: {args.mona_bottleneck}")
    else:
        inject_lora_to_biomedclip(model, lora_r=args.lora_r, lora_alpha=args.lora_alpha, lora_dropout=args.lora_dropout,
                                  num_layers=args.lora_layers, tune_text_encoder=args.tune_text_encoder)
        key = "lora"
        logging.info(f"LoRA rank: {args.lora_r}, alpha: {args.lora_alpha}, dropout: {args.lora_dropout}")
    for name, p in model.named_parameters():
        if key in name.lower():
            p.requires_grad = True
    model.to(args.device)
    tr = sum(p.numel() for p in model.parameters() if p.requires_grad)
    tot = sum(p.numel() for p in model.parameters())
    logging.info(f"Trainable parameters: {tr:,} / {tot:,} ({100 * tr / tot:.2f}%)")
    return model, tokenizer


def _save_checkpoint(model, args, save_path):
    if args.method == "full":                                    # reference :200-208: the whole state dict
        torch.save(model.state_dict(), save_path)
        return
    key = "mona" if args.method == "mona" else "lora"
    torch.save({n: p.data.clone() for n, p in model.named_parameters() if key in n.lower()}, save_path)


def train(args):
    """The loop of reference :211-361 on the measured step: every loader batch goes through engine.ContrastiveLoop.micro (contrastive_micro: text tower on its
    own stream, image tower in two slices, three-byte residual gradients; guarded accumulate) and every accumulation boundary through the device-guarded
    clip + AdamW — the same functions bench.py times.  The host reads nothing per batch: the non-finite skip (:281-285) is decided on the device, the epoch's
    loss sum / counts / skipped indices come back in ONE read at the end of the epoch, batches arrive through a double-buffered prefetcher."""
    rank, _, world = dist_env()
    if not torch.cuda.is_initialized():                         # a fresh CLI process (not a caller that is already running other GPU / CPU work in this process)
        torch.set_num_threads(max(1, min(4, torch.get_num_threads())))      # host tensor work here is one staging copy per batch; the default (every logical CPU of the node) oversubscribes a job's CPU share
    dm = dataset_finetune.DataModule(args, rank=rank, world=world, tokenizer=make_tokenizer(args))
    trainloader, valloader = dm.train_dataloader(), dm.val_dataloader()
    dm.start_workers()                                         # loader worker processes are forked BEFORE this process touches the GPU
    bind_device(args)                                          # data parallel: cuda:LOCAL_RANK before anything is allocated
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    UF.set_dropout_seed(args.seed + 7919 * rank)                # ranks draw different dropout masks, like different micro-batches
    model, tokenizer = prepare_model(args)
    model.train()
    logging.info(model_summary({"model": model}))
    criterion = InfoNCELoss(temperature=args.temperature)
    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=args.lr,
                               betas=(args.beta1_adam, args.beta2_adam), weight_decay=args.weight_decay, max_norm=args.grad_clip)
    if world > 1:
        init_data_parallel(opt)
    updates_per_epoch = math.ceil(len(trainloader) / args.accumulation_steps)
    total_updates = updates_per_epoch * args.epochs
    logging.info(f"Gradient accumulation steps: {args.accumulation_steps}; updates per epoch: {updates_per_epoch}; world: {world}")
    loop = ContrastiveLoop(model, criterion, opt, accumulation_steps=args.accumulation_steps, lr=args.lr, lr_min=args.lr_min, total_updates=total_updates)
    train_pf = DevicePrefetcher(trainloader, tokenizer, args.device)
    val_pf = DevicePrefetcher(valloader, tokenizer, args.device)

    best_loss, best_epoch, patience = float("inf"), 0, 0
    avg_train, update_count, epoch_ms = 0.0, 0, []
    dm.set_epoch(0)
    batches = iter(train_pf)                                    # the producer starts now: the first batches are resident when the epoch begins
    for epoch in range(args.epochs):
        model.train()
        loop.begin_epoch(len(trainloader))
        torch.cuda.synchronize()
        t0, w0, enq, gw0 = time.perf_counter(), train_pf.wait_s, 0.0, getattr(opt, "gpu_wait_s", 0.0) + UF._STATE.get("gpu_wait_s", 0.0)
        for batch_idx, (images, tokens, ready) in enumerate(batches):
            t1 = time.perf_counter()
            loop.micro(images, tokens, batch_idx, ready=ready)
            enq += time.perf_counter() - t1
        t2 = time.perf_counter()
        g = loop.end_epoch()                                     # the epoch's one host read (reference: loss.item() per batch, :290)
        epoch_ms.append({"ms": (time.perf_counter() - t0) * 1e3, "updates": g["updates"] - update_count, "batches": len(trainloader),
                         "loader_wait_ms": (train_pf.wait_s - w0) * 1e3, "enqueue_ms": enq * 1e3, "drain_ms": (time.perf_counter() - t2) * 1e3,
                         "host_waited_for_gpu_ms": (getattr(opt, "gpu_wait_s", 0.0) + UF._STATE.get("gpu_wait_s", 0.0) - gw0) * 1e3})
        update_count = g["updates"]
        for i in g["skipped_batches"]:
            logging.warning(f"Non-finite loss detected at batch {i} in epoch {epoch + 1}, skipping batch")
        ep_loss, ep_n = g["loss_sum"], g["epoch_accumulated"]
        if epoch + 1 < args.epochs:                              # next epoch's first batches load while this epoch validates
            dm.set_epoch(epoch + 1)
            batches = iter(train_pf)
        model.eval()
        vstat = torch.zeros(2, device=args.device)
        with torch.no_grad():
            for images, tokens, ready in val_pf:
                torch.cuda.current_stream().wait_event(ready)
                loss = criterion(model.encode_image(images), model.encode_text(tokens))
                ok = torch.isfinite(loss)
                vstat += torch.stack([torch.where(ok, loss, torch.zeros_like(loss)), ok.to(loss.dtype)])
        val_loss, val_n = vstat.tolist()
        # every rank evaluated its own shard: the epoch's figures are the sums over ranks, so that the checkpoint / patience /
        # early-stop decisions below are identical on every rank (a rank that stopped alone would strand the others)
        val_loss, val_n, ep_loss, ep_n = sum_over_ranks(val_loss, val_n, ep_loss, ep_n)
        avg_val, avg_train = (val_loss / val_n if val_n else 0.0), (ep_loss / ep_n if ep_n else 0.0)
        if avg_val < best_loss:
            best_loss, patience, best_epoch = avg_val, 0, epoch
            if rank == 0:
                _save_checkpoint(model, args, os.path.join(args.train_snapshot_path, "best_model.pth"))
            logging.info(f"\nBest model saved at epoch {epoch + 1} with validation loss {best_loss:.4f}")
        else:
            patience += 1
        logging.info(f"Epoch {epoch + 1}: Train={avg_train:.4f}, Val={avg_val:.4f}, Best={best_loss:.4f}")
        if patience >= args.patience:
            logging.info(f"\nEarly stopping at epoch {epoch + 1} as validation loss did not improve for {args.patience} epochs.")
            break
    logging.info(f"\n✓ Training completed! Best validation loss: {best_loss:.4f} at epoch {best_epoch + 1}")
    train_pf.close()                                           # an epoch prefetched ahead and abandoned by early stopping
    val_pf.close()
    dm.shutdown()
    if world > 1:
        from uia_hip import ops
        import torch.distributed as dist
        dist.barrier()
        ops.comm_destroy()
    return {"best_val": best_loss, "updates": update_count, "last_train": avg_train, "rank": rank, "world": world, "epochs": epoch_ms}


def main(argv=None):
    args = get_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.train_snapshot_path = f"runs/{args.exp}"
    os.makedirs(args.train_snapshot_path, exist_ok=True)
    setup_logging(args, args.train_snapshot_path)
    out = train(args)
    if args.stats_json and out.get("rank", 0) == 0:
        import json
        with open(args.stats_json, "w") as f:
            json.dump(out, f)
    return out


if __name__ == "__main__":
    main()
