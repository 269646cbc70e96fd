"""MetaCLIP contrastive fine-tuning with Mona adapters on the MI355X HIP path — drop-in for
/root/reference/src/models/metaclip/finetune.py (the clip/ and unimedclip/ entry points of the reference share this loop).

Same command line (every flag and default of reference :26-64; default --mona_variant noise_aware, batch 64, no gradient
accumulation) and loop semantics (:92-218): per batch InfoNCE on L2-normalised features, non-finite batches skipped,
clip_grad_norm_(1.0) → AdamW → cosine LR every iteration, a validation pass per epoch in eval mode, best-val checkpoint of
the parameters whose name contains "mona", early stopping by --patience, runs/<exp>/{log.log,best_model.pth}.
Added, non-breaking: --dtype, --synthetic / --data_pt, --ckpt_path, --model_config, data parallelism under
torch.distributed.run.  The post-training zero-shot subprocess (:220-268) is outside the hot path and not launched.
"""
import argparse
import logging
import os
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))

import numpy as np
import torch

from src.adapters import inject_mona_variant_to_open_clip
from src.datasets import finetune as dataset_finetune
from src.losses import InfoNCELoss
from src.third_party.open_clip.model import SyntheticClipTokenizer, create_metaclip
from src.utils.tools import default_device, model_summary, parse_config, setup_logging
from uia_hip import functional as UF
from uia_hip.engine import ContrastiveLoop, DevicePrefetcher, FlatAdapterOptimizer, bind_device, dist_env, init_data_parallel, sum_over_ranks


def get_args(argv=None):
    p = argparse.ArgumentParser("MetaCLIP Fine-tuning with Frequency-Enhanced MONA")
    p.add_argument("--img_size", type=int, default=224)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--strong_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--weak_augs", default=False, action=argparse.BooleanOptionalAction)
    p.add_argument("--exp", type=str, default="metaclip_finetune")
    p.add_argument("--in_channels", type=int, default=3)
    p.add_argument("--mona_variant", type=str, default="noise_aware")
    p.add_argument("--mona_bottleneck", type=int, default=64)
    p.add_argument("--mona_layers", type=int, default=None)
    p.add_argument("--temperature", type=float, default=0.07)
    p.add_argument("--uniformity_weight", type=float, default=0)
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
    p.add_argument("--ckpt_path", type=str, default=None, help="open_clip state dict (.pt); random init if absent")
    p.add_argument("--model_config", type=str, default=None, help="python dict literal overriding the model geometry (tests)")
    return p.parse_args(argv)


def make_tokenizer(args):
    tc = ((parse_config(args.model_config) if args.model_config else None) or {}).get("text_cfg", {})
    return SyntheticClipTokenizer(tc.get("context_length", 77), tc.get("vocab_size", 49408))


def prepare_model(args):
    cfg = parse_config(args.model_config) if args.model_config else None
    state = torch.load(args.ckpt_path, map_location="cpu") if args.ckpt_path else None
    model = create_metaclip(state_dict=state, config=cfg, seed=args.seed)
    tokenizer = make_tokenizer(args)
    for p in model.parameters():
        p.requires_grad = False
    model, mona_count = inject_mona_variant_to_open_clip(model, variant=args.mona_variant, bottleneck_dim=args.mona_bottleneck,
                                                         num_layers=args.mona_layers)
    for name, p in model.named_parameters():
        if "mona" in name.lower():
            p.requires_grad = True
    model.float()
    model.to(args.device)
    return model, tokenizer


def _normalise(fi, ft):
    """reference metaclip/finetune.py:147-148: the entry point L2-normalises both feature matrices before the loss"""
    return fi / fi.norm(dim=-1, keepdim=True), ft / ft.norm(dim=-1, keepdim=True)


def train(args, prepare=None, tokenizer_of=None):
    """prepare / tokenizer_of: another model family's `prepare_model(args) -> (model, tokenizer)` / `make_tokenizer(args)` around the same loop (the reference's clip/ and
    unimedclip/ entry points repeat it verbatim: src/models/clip/finetune.py:92-215).

    reference metaclip/finetune.py:98-215 on the measured step (engine.ContrastiveLoop with one micro-batch per update, the reference has no accumulation here): no host
    read per batch; the non-finite skip (:153-155: no backward, no optimiser step, no scheduler step, iter_num not advanced) is the device-guarded update."""
    rank, _, world = dist_env()
    if not torch.cuda.is_initialized():                         # a fresh CLI process (not a caller that is already running other GPU / CPU work in this process)
        torch.set_num_threads(max(1, min(4, torch.get_num_threads())))      # host tensor work here is one staging copy per batch; the default (every logical CPU of the node) oversubscribes a job's CPU share
    dm = dataset_finetune.DataModule(args, rank=rank, world=world, tokenizer=(tokenizer_of or make_tokenizer)(args))
    trainloader, valloader = dm.train_dataloader(), dm.val_dataloader()
    dm.start_workers()                                         # loader worker processes are forked BEFORE this process touches the GPU
    bind_device(args)                                          # data parallel: cuda:LOCAL_RANK before anything is allocated
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    UF.set_dropout_seed(args.seed + 7919 * rank)
    model, tokenizer = (prepare or prepare_model)(args)
    model.train()
    logging.info(model_summary({"model": model}))
    criterion = InfoNCELoss(temperature=args.temperature)
    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=args.lr,
                               betas=(args.beta1_adam, args.beta2_adam), weight_decay=args.weight_decay, max_norm=1.0)
    if world > 1:
        init_data_parallel(opt)
    max_iters = len(trainloader) * args.epochs
    loop = ContrastiveLoop(model, criterion, opt, accumulation_steps=1, lr=args.lr, lr_min=args.lr_min, total_updates=max_iters, features=_normalise, discard_on_skip=True)      # zero_grad() before every backward in the reference (:160)
    train_pf = DevicePrefetcher(trainloader, tokenizer, args.device)
    val_pf = DevicePrefetcher(valloader, tokenizer, args.device)
    iter_num, best_loss, best_epoch, patience = 0, float("inf"), 0, 0
    train_loss, epoch_ms = 0.0, []
    dm.set_epoch(0)
    batches = iter(train_pf)
    for epoch in range(args.epochs):
        model.train()
        loop.begin_epoch(len(trainloader))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for batch_idx, (images, tokens, ready) in enumerate(batches):
            loop.micro(images, tokens, batch_idx, ready=ready)
        g = loop.end_epoch()                                     # the epoch's one host read
        epoch_ms.append({"ms": (time.perf_counter() - t0) * 1e3, "updates": g["updates"] - iter_num, "batches": len(trainloader)})
        for _ in g["skipped_batches"]:
            logging.warning(f"Non-finite loss detected at iteration {iter_num}, skipping batch")
        iter_num = g["updates"]
        train_loss = g["loss_sum"] / max(1, len(trainloader))
        if epoch + 1 < args.epochs:
            dm.set_epoch(epoch + 1)
            batches = iter(train_pf)
        model.eval()
        vsum = torch.zeros((), device=args.device)
        with torch.no_grad():
            for images, tokens, ready in val_pf:
                torch.cuda.current_stream().wait_event(ready)
                loss = criterion(*_normalise(model.encode_image(images), model.encode_text(tokens)))
                vsum += torch.where(torch.isfinite(loss), loss, torch.zeros_like(loss))
        val_loss = float(vsum) / max(1, len(valloader))
        val_loss, train_loss = (v / world for v in sum_over_ranks(val_loss, train_loss))      # same figures, same decisions on every rank
        logging.info(f"Epoch {epoch + 1}/{args.epochs}: Train={train_loss:.4f}, Val={val_loss:.4f}, Best={best_loss:.4f}")
        if val_loss < best_loss:
            best_loss, best_epoch, patience = val_loss, epoch, 0
            if rank == 0:
                torch.save({n: p.data.clone() for n, p in model.named_parameters() if "mona" in n.lower()},
                           os.path.join(args.train_snapshot_path, "best_model.pth"))
            logging.info(f"Best model saved at epoch {epoch + 1}")
        else:
            patience += 1
        if patience >= args.patience:
            logging.info(f"Early stopping triggered at epoch {epoch + 1}")
            break
    logging.info(f"\n✓ Training completed! Best loss: {best_loss:.4f} (epoch {best_epoch + 1})")
    train_pf.close()                                           # an epoch prefetched ahead and abandoned by early stopping
    val_pf.close()
    dm.shutdown()
    if world > 1:
        from uia_hip import ops
        import torch.distributed as dist
        dist.barrier()
        ops.comm_destroy()
    return {"best_val": best_loss, "iters": iter_num, "last_train": train_loss, "epochs": epoch_ms}


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
