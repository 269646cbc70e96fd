"""BiomedCLIP segmentation with the feature-pyramid adapter on the HIP path — counterpart of
/root/reference/src/models/biomedclip/segmentation.py.

Kept: the command line (:28-72; default --mona_variant hybrid, batch 32, AdamW betas 0.9/0.95, no gradient clipping), model
assembly (:81-135: optional LoRA or Mona adapters loaded BY NAME from a fine-tune checkpoint, TimmCLIPAdapter(task="seg") on
layers 3/6/9, freeze_clip_backbone()), the loop (:138-255): DiceCE(to_onehot_y, softmax, squared_pred) per iteration with a
per-iteration cosine schedule, validation by mean foreground Dice every 10 epochs and at the last one, early stopping by
--patience, and the checkpoint dict {"reduces", "blocks", "seg_head", "mona"} (:212-225) under runs/<exp>/<dataset>/train.
the test pass after every validation (:259-280), `test()` (:283-355: checkpoint -> heads + Mona parameters by name -> metrics over the test split -> the
Metric / Mean / Std table, results.csv and the backup folder) and `main` (train unless --test, then ALWAYS test).
Different on purpose: the iteration is engine.segmentation_step with nothing read on the host, batches through engine.DevicePrefetcher (one grayscale channel +
a uint8 mask per image, repeated on the device), Dice / IoU on the device; MONAI's HD95 / ASD (scipy surface distances on the host) are reported as NaN and
TensorBoard scalars go to <train dir>/log/scalars.jsonl.  Data: `--synthetic` or `--data_pt` (src/datasets/segmentation.py).
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

from src.adapters import inject_lora_to_biomedclip, inject_mona_variant_to_open_clip
from src.losses.dice import DiceCELoss
from src.models.biomedclip.zero_shot import load_adapter_by_name
from src.datasets import segmentation as dataset_seg
from src.datasets.segmentation import as_model_input
from src.third_party.biomedclip.model import create_biomedclip
from src.third_party.timm.clip_adapter import TimmCLIPAdapter
from src.utils.tools import MetricAccumulator, ScalarLog, default_device, fresh_viz_dir, model_summary, parse_config, report_test, setup_logging
from uia_hip import functional as UF
from uia_hip.engine import DevicePrefetcher, FlatAdapterOptimizer, bind_device, cosine_lr, dist_env, init_data_parallel, segmentation_step


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
    p.add_argument("--device", type=str, default=default_device())
    p.add_argument("--patience", type=int, default=15)
    p.add_argument("--test", default=False, action="store_true")
    # additions of this build
    p.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_train", type=int, default=256)
    p.add_argument("--synthetic_val", type=int, default=64)
    p.add_argument("--synthetic_test", type=int, default=64)
    p.add_argument("--stats_json", type=str, default=None)
    p.add_argument("--data_pt", type=str, default=None)
    p.add_argument("--ckpt_path", type=str, default=None, help="open_clip BiomedCLIP state dict (.pt); random init if absent")
    p.add_argument("--model_config", type=str, default=None)
    p.add_argument("--extract_layers", type=str, default="3,6,9", help="transformer blocks tapped by the adapter (reference: fixed 3,6,9)")
    p.add_argument("--val_every", type=int, default=10, help="epochs between validations (reference: fixed 10)")
    return p.parse_args(argv)


criterion = DiceCELoss(smooth_nr=1e-8, smooth_dr=1e-8)            # reference :76


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


def evaluate(model, loader_pf, args, accumulator):
    cur = torch.cuda.current_stream()
    with torch.no_grad():
        for images, labels, ready in loader_pf:
            cur.wait_event(ready)
            images, labels = as_model_input(images, labels, args.in_channels, widen=False)
            accumulator.update(model(images).detach(), labels.detach())


def train(args):
    rank, _, world = dist_env()
    if not torch.cuda.is_initialized():
        torch.set_num_threads(max(1, min(4, torch.get_num_threads())))
    dm = dataset_seg.DataModule(args, rank=rank, world=world)
    trainloader, valloader, testloader = dm.train_dataloader(), dm.val_dataloader(), dm.test_dataloader()
    dm.start_workers()                                         # loader worker processes are forked BEFORE this process touches the GPU
    bind_device(args)                                          # data parallel: cuda:LOCAL_RANK before anything is allocated
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    UF.set_dropout_seed(args.seed + 7919 * rank)
    model = prepare_model(args)
    model.train()
    logging.info(model_summary({"model": model}))
    writer = ScalarLog(args.train_snapshot_path + "/log")
    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=args.lr, betas=(args.beta1, args.beta2),
                               weight_decay=args.weight_decay, max_norm=0.0)
    if world > 1:
        init_data_parallel(opt)
    max_iters = len(trainloader) * args.epochs
    train_pf = DevicePrefetcher(trainloader, None, args.device, second=dataset_seg.second_of)
    val_pf = DevicePrefetcher(valloader, None, args.device, second=dataset_seg.second_of)
    test_pf = DevicePrefetcher(testloader, None, args.device, second=dataset_seg.second_of)
    iter_num, best_val_dice, patience, logged, last = 0, 0.0, 0, [], None
    epoch_ms = []
    cur = torch.cuda.current_stream()
    bufs = {}                                                    # the widened training batch lives in the same two tensors every iteration
    dm.set_epoch(0)
    batches = iter(train_pf)
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0, n_it = time.perf_counter(), 0
        for images, labels, ready in batches:
            cur.wait_event(ready)
            images, labels = as_model_input(images, labels, args.in_channels, bufs, widen=False)
            loss, _ = segmentation_step(model, criterion, opt, images, labels, lr=cosine_lr(args.lr, args.lr_min, iter_num, max_iters))
            if iter_num % 10 == 0:
                logged.append((iter_num, loss))
            iter_num += 1
            n_it += 1
        torch.cuda.synchronize()
        epoch_ms.append({"ms": (time.perf_counter() - t0) * 1e3, "updates": n_it})
        validate = (epoch > 0 and epoch % args.val_every == 0) or (epoch == args.epochs - 1)
        if epoch + 1 < args.epochs:
            dm.set_epoch(epoch + 1)
            batches = iter(train_pf)
        if not validate:
            continue
        model.eval()
        for it, l in logged:
            writer.add_scalar(f"{args.exp}/train_loss", l, it)
        last = float(logged[-1][1]) if logged else last
        logged = []
        accumulator = MetricAccumulator(type="seg", criterion=criterion, num_classes=args.num_classes)
        evaluate(model, val_pf, args, accumulator)
        stats = accumulator.compute()
        accumulator.reset()
        for k in ("loss", "dice_mean", "iou_mean"):
            writer.add_scalar(f"{args.exp}/val_{k.replace('_mean', '')}", stats[k], iter_num)
        writer.flush()
        if stats["dice_mean"] > best_val_dice:
            patience, best_val_dice = 0, stats["dice_mean"]
            if rank == 0:
                torch.save(checkpoint_dict(model), os.path.join(args.train_snapshot_path, "best_model.pth"))
        else:
            patience += 1
        if patience >= args.patience:
            logging.info(f"\nEarly stopping at epoch {epoch + 1}")
            break
        logging.info(f"\titer: {iter_num}, loss: {stats['loss']:.4f}, dice: {stats['dice_mean'] * 100:.2f}, "
                     f"iou: {stats['iou_mean'] * 100:.2f}, hd95: {stats['hd95_mean']:.2f}, asd: {stats['asd_mean']:.2f}")
        evaluate(model, test_pf, args, accumulator)             # reference :259-280
        tstats = accumulator.compute()
        accumulator.reset()
        for k in ("loss", "dice_mean", "iou_mean"):
            writer.add_scalar(f"{args.exp}/test_{k.replace('_mean', '')}", tstats[k], iter_num)
        writer.flush()
        model.train()
    writer.close()
    for pf in (train_pf, val_pf, test_pf):
        pf.close()
    dm.shutdown()
    if world > 1:
        from uia_hip import ops
        import torch.distributed as dist
        dist.barrier()
        ops.comm_destroy()
    if rank == 0 and not os.path.exists(os.path.join(args.train_snapshot_path, "best_model.pth")):
        logging.warning("no validation improved on a mean Dice of 0.0: saving the last iterate as best_model.pth (the reference has no checkpoint then and fails in test())")
        torch.save(checkpoint_dict(model), os.path.join(args.train_snapshot_path, "best_model.pth"))
    out = {"iters": iter_num, "best_val_dice": best_val_dice, "last_loss": last, "rank": rank, "world": world, "epochs": epoch_ms}
    if args.stats_json and rank == 0:
        import json
        with open(args.stats_json, "w") as f:
            json.dump(out, f)
    return out


@torch.no_grad()
def test(args):
    """reference :283-355."""
    logging.info("Start testing")
    rank, _, _ = dist_env()
    dm = dataset_seg.DataModule(args, rank=0, world=1)          # every rank evaluates the whole split
    testloader = dm.test_dataloader()
    dm.start_workers()
    bind_device(args)
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model = prepare_model(args)
    saved_best = os.path.join(args.train_snapshot_path, "best_model.pth")
    adapter_state_dict = torch.load(saved_best, map_location="cpu")
    model.reduces.load_state_dict(adapter_state_dict["reduces"])
    model.blocks.load_state_dict(adapter_state_dict["blocks"])
    model.seg_head.load_state_dict(adapter_state_dict["seg_head"])
    mona_state_dict = adapter_state_dict["mona"]
    with torch.no_grad():
        for name, param in model.named_parameters():            # :297-300 (copy_ instead of re-pointing .data: the T copies of the weights are keyed by version)
            if "mona" in name:
                param.copy_(mona_state_dict[name].to(param.device))
    model.eval()
    if rank == 0:
        fresh_viz_dir(args)
    accumulator = MetricAccumulator(type="seg", criterion=criterion, num_classes=args.num_classes)
    test_pf = DevicePrefetcher(testloader, None, args.device, second=dataset_seg.second_of)
    evaluate(model, test_pf, args, accumulator)
    stats = accumulator.compute()
    accumulator.reset()
    test_pf.close()
    dm.shutdown()
    stats["results_csv"] = report_test(args, stats, saved_best, rank)
    return stats


def main(argv=None):
    args = get_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    args.train_snapshot_path = f"runs/{args.exp}/{args.dataset}/train"
    args.test_snapshot_path = f"runs/{args.exp}/{args.dataset}/test"
    for path in (args.train_snapshot_path, args.test_snapshot_path):
        os.makedirs(path, exist_ok=True)
    out = {}
    if not args.test:
        setup_logging(args, args.train_snapshot_path)
        out = train(args)
    setup_logging(args, args.test_snapshot_path)
    out["test"] = test(args)
    return out


if __name__ == "__main__":
    main()
