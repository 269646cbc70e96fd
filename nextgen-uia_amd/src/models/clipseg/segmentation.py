"""CLIPSeg segmentation on the HIP path — drop-in for /root/reference/src/models/clipseg/segmentation.py (BASELINE configs[3]).

Kept, by reference line:
  :28-66    every flag with its default (--dataset LN-INT, --epochs 1000, --lr_min 1e-8, --beta1 0.9 / --beta2 0.95, --patience 15, --strong_augs / --weak_augs
            as BooleanOptionalAction, --version, --ckpt, --patch_size, --test ...)
  :69-82    get_prompt: LN-INT / LN-EXT -> ln_prompt, BUSI -> busi_prompt, DDTI / TN3K -> thyroid_prompt, Prostate -> prostate_prompt, anything else None
  :85       DiceCELoss(to_onehot_y, softmax, squared_pred, smooth 1e-8 / 1e-8), a module-level `criterion`
  :88-105   prepare_model: CLIP from --ckpt in fp32 -> CLIPSegAdapter -> freeze_clip_backbone()
  :108-236  train: AdamW(decoder, lr, betas, weight_decay), CosineAnnealingLR over len(trainloader) * epochs ITERATIONS stepped per iteration, the fixed prompt
            repeated over the batch (:142), validation when (epoch > 0 and epoch % 10 == 0) or at the last epoch -> MetricAccumulator -> best mean Dice ->
            {"decoder": state_dict} in runs/<exp>/<dataset>/train/best_model.pth (:190-197), else patience += 1, early stop at --patience (:201-204), then the
            same pass over the test split (:212-231)
  :238-308  test: best_model.pth -> decoder, metrics over the test split, a Metric / Mean / Std table logged and written to
            runs/<exp>/<dataset>/test/<time>_iou=<iou>/results.csv beside copies of the checkpoint and the log
  :311-341  main: seeds, the two run directories, train unless --test, then ALWAYS test.

Different on purpose: the iteration is engine.segmentation_step — the step `bench.py --config clipseg` times — and nothing is read on the host per iteration
(the reference's loss.item() every tenth iteration goes to a log of device scalars that is flushed at validation time); batches arrive through
engine.DevicePrefetcher (loader workers -> shared-memory ring -> copy stream), one grayscale channel + a uint8 mask per image, repeated to three channels on the
device; validation metrics are computed on the device (Dice / IoU, MONAI semantics; HD95 / ASD are host-side scipy work and are reported as NaN); TensorBoard
scalars go to <train dir>/log/scalars.jsonl.  Data: --synthetic or --data_pt (src/datasets/segmentation.py).  Build-only flags are listed after the
reference's in get_args.  Data parallel under torch.distributed.run: every rank trains on its shard, ONE all-reduce of the decoder gradients per iteration.
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

from src.datasets import segmentation as dataset_seg
from src.datasets.segmentation import as_model_input, synthetic_batch      # noqa: F401  (synthetic_batch: bench.py and tools import it from here)
from src.losses.dice import DiceCELoss
from src.models.clipseg.prompt import busi_prompt, ln_prompt, prostate_prompt, thyroid_prompt
from src.third_party.openai_clip.clipseg_adapter import CLIPSegAdapter
from src.third_party.openai_clip.model import CLIP, build_model
from src.utils.tools import MetricAccumulator, ScalarLog, default_device, fresh_viz_dir, model_summary, report_test, setup_logging
from uia_hip import functional as UF
from uia_hip.engine import DevicePrefetcher, FlatAdapterOptimizer, bind_device, cosine_lr, dist_env, init_data_parallel, segmentation_step


def get_args(argv=None):
    """Get arguments from command line (reference :28-66, then this build's additions)."""
    parser = argparse.ArgumentParser("Adaptation of Visual Foundation Model for Medical Ultrasound Image Analysis")
    # Data related
    parser.add_argument("--exp", type=str, default="clipseg")
    parser.add_argument("--dataset", type=str, default="LN-INT", help="Dataset name")
    parser.add_argument("--img_size", type=int, default=224, help="Image width and height")
    parser.add_argument("--patch_size", type=int, default=16, help="Patch size")
    parser.add_argument("--num_workers", type=int, default=8)
    # Augmentation related (accepted; augmentation belongs to the reference's PIL loaders)
    parser.add_argument("--strong_augs", default=True, action=argparse.BooleanOptionalAction, help="Use strong augs")
    parser.add_argument("--weak_augs", default=True, action=argparse.BooleanOptionalAction, help="Use weak augs")
    # Model related
    parser.add_argument("--version", type=str, default="ViT-B/16")
    parser.add_argument("--ckpt", type=str, default="ckpt/ViT-B-16.pt")
    parser.add_argument("--in_channels", type=int, default=3)
    parser.add_argument("--num_classes", type=int, default=2)
    parser.add_argument("--reduce_dim", type=int, default=512)           # unused by the reference as well (the decoder's own 64 applies)
    # Training related
    parser.add_argument("--seed", type=int, default=1)
    parser.add_argument("--epochs", type=int, default=1000)
    parser.add_argument("--batch_size", type=int, default=32)
    parser.add_argument("--lr", type=float, default=1e-4)
    parser.add_argument("--lr_min", type=float, default=1e-8)
    parser.add_argument("--weight_decay", type=float, default=0.01)
    parser.add_argument("--beta1", type=float, default=0.9)
    parser.add_argument("--beta2", type=float, default=0.95)
    parser.add_argument("--device", type=str, default=default_device())
    parser.add_argument("--patience", type=int, default=15, help="Early stopping patience (10 * N epochs)")
    # Testing related
    parser.add_argument("--test", default=False, action="store_true", help="Load local checkpoint for testing")
    # ---- additions of this build (none changes a reference default)
    parser.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"], help="operand precision of the HIP kernels (masters stay fp32)")
    parser.add_argument("--synthetic", action="store_true", help="synthetic images + ellipse masks instead of ../data/NextGen-UIA")
    parser.add_argument("--synthetic_train", type=int, default=256)
    parser.add_argument("--synthetic_val", type=int, default=64)
    parser.add_argument("--synthetic_test", type=int, default=64)
    parser.add_argument("--data_pt", type=str, default=None, help=".pt file {images, labels[, names, split]} (src/datasets/segmentation.py)")
    parser.add_argument("--decoder_ckpt", type=str, default=None, help="CIDAS/clipseg-rd64-refined decoder state dict (the reference downloads it)")
    parser.add_argument("--val_every", type=int, default=10, help="epochs between validations (reference: fixed 10)")
    parser.add_argument("--stats_json", type=str, default=None, help="write per-epoch timings / counters of the training loop here")
    return parser.parse_args(argv)


def get_prompt(args):
    """Get tokenized prompt according to the dataset (reference :69-82)."""
    prompt = None
    if args.dataset == "LN-INT" or args.dataset == "LN-EXT":
        prompt = ln_prompt
    elif args.dataset == "BUSI":
        prompt = busi_prompt
    elif args.dataset == "DDTI" or args.dataset == "TN3K":
        prompt = thyroid_prompt
    elif args.dataset == "Prostate":
        prompt = prostate_prompt
    return prompt


# Loss function (reference :85)
criterion = DiceCELoss(smooth_nr=1e-8, smooth_dr=1e-8)

_GEOMETRY = {   # random-init geometry per --version when --synthetic runs without a checkpoint: (embed, layers, width, patch, text width, text heads)
    "ViT-B/32": (512, 12, 768, 32, 512, 8), "ViT-B/16": (512, 12, 768, 16, 512, 8), "ViT-L/14": (768, 24, 1024, 14, 768, 12),
}


def load_clip(args):
    """clip.load(args.ckpt) of the reference (:90; clip.py:97-150 accepts a TorchScript archive or a plain state dict)."""
    if args.ckpt and os.path.exists(args.ckpt):
        try:
            state = torch.jit.load(args.ckpt, map_location="cpu").state_dict()
        except RuntimeError:
            state = torch.load(args.ckpt, map_location="cpu")
            state = state.get("state_dict", state) if isinstance(state, dict) else state.state_dict()
        return build_model(state)
    if not args.synthetic:
        raise FileNotFoundError(f"Model {args.ckpt} not found (download the OpenAI {args.version} checkpoint, or pass --synthetic for randomly initialised weights)")
    if args.version not in _GEOMETRY:
        raise NotImplementedError(f"--version {args.version}: only ViT variants run on this path ({', '.join(_GEOMETRY)})")
    e, layers, width, patch, tw, th = _GEOMETRY[args.version]
    logging.info(f"{args.ckpt} not found: randomly initialised {args.version} (--synthetic)")
    torch.manual_seed(args.seed)
    return CLIP(e, args.img_size, layers, width, patch, 77, 49408, tw, th, 12)


def prepare_model(args):
    clip_model = load_clip(args)
    clip_model.float()                                           # reference :92-95
    adapter = CLIPSegAdapter(clip_model=clip_model)
    if args.decoder_ckpt:
        adapter.decoder.load_state_dict(torch.load(args.decoder_ckpt, map_location="cpu"))
    adapter.to(args.device)
    adapter.freeze_clip_backbone()
    return adapter


def _batch_prompt(prompt, cache, n):
    """prompt.repeat(n, 1) (:142), ONE tensor object per batch size: CLIPSegAdapter recognises an unchanged prompt tensor without looking at its contents."""
    if n not in cache:
        cache[n] = prompt.repeat(n, 1)
    return cache[n]


def evaluate(model, loader_pf, prompt, cache, args, accumulator):
    """One pass of a validation / test split (:157-164, :212-218); returns the last batch (the reference logs its first four images)."""
    cur = torch.cuda.current_stream()
    last = None
    with torch.no_grad():
        for images, labels, ready in loader_pf:
            cur.wait_event(ready)
            images, labels = as_model_input(images, labels, args.in_channels, widen=False)
            preds = model(images, input_ids=_batch_prompt(prompt, cache, images.shape[0]))
            accumulator.update(preds.detach(), labels.detach())
            last = (images, labels, preds)
    return last


def train(args):
    rank, _, world = dist_env()
    if not torch.cuda.is_initialized():
        torch.set_num_threads(max(1, min(4, torch.get_num_threads())))
    prompt = get_prompt(args)
    if prompt is None:
        raise ValueError(f"no prompt for --dataset {args.dataset} (LN-INT, LN-EXT, BUSI, DDTI, TN3K, Prostate; reference get_prompt returns None and fails at .repeat)")
    dm = dataset_seg.DataModule(args, rank=rank, world=world)
    trainloader, valloader, testloader = dm.train_dataloader(), dm.val_dataloader(), dm.test_dataloader()
    dm.start_workers()                                          # loader worker processes are forked BEFORE this process touches the GPU
    bind_device(args)
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model = prepare_model(args)
    model.train()
    logging.info(model_summary({"model": model}))
    writer = ScalarLog(args.train_snapshot_path + "/log")
    logging.info("Start training")
    prompt = prompt.to(args.device)
    cache = {}

    opt = FlatAdapterOptimizer([(n, p) for n, p in model.named_parameters() if p.requires_grad], lr=args.lr, betas=(args.beta1, args.beta2),
                               weight_decay=args.weight_decay, max_norm=0.0)                                  # :121-126; no clipping in this loop
    if world > 1:
        init_data_parallel(opt)
    max_epoch = args.epochs
    max_iters = len(trainloader) * max_epoch                    # :128-130
    train_pf = DevicePrefetcher(trainloader, None, args.device, second=dataset_seg.second_of)
    val_pf = DevicePrefetcher(valloader, None, args.device, second=dataset_seg.second_of)
    test_pf = DevicePrefetcher(testloader, None, args.device, second=dataset_seg.second_of)

    iter_num, best_val_dice, patience_counter = 0, 0.0, 0
    epoch_ms, logged = [], []
    cur = torch.cuda.current_stream()
    bufs = {}                                                    # the widened training batch lives in the same two tensors every iteration
    dm.set_epoch(0)
    batches = iter(train_pf)
    ab = os.environ.get("UIA_SEG_AB", "")                     # measurement knobs (tools/ab_clipseg_entry.sh); none is set in normal use
    if "nogc" in ab:
        import gc
        gc.disable()
    if "switch" in ab:
        sys.setswitchinterval(0.05)
    sampler = None
    if "power" in ab:                                            # board power / shader clock during the epochs (uia_hip.telemetry: amdgpu hwmon files, no HIP call)
        from uia_hip.telemetry import PowerSampler
        sampler = PowerSampler(torch, torch.device(args.device))
        sampler.start()
    for epoch in range(max_epoch):
        torch.cuda.synchronize()
        t0, w0, enq = time.perf_counter(), train_pf.wait_s, 0.0
        n_it = 0
        if "resident" in ab and epoch > 0:                      # the same device batch every iteration, the prefetcher idle: the loop's floor
            if epoch == 1:
                first = next(iter(train_pf))
                train_pf.close()
            batches = (first for _ in range(len(trainloader)))
        for images, labels, ready in batches:
            t1 = time.perf_counter()
            cur.wait_event(ready)
            images, labels = as_model_input(images, labels, args.in_channels, bufs, widen=False)
            # scheduler.step() follows optimizer.step() (:147-148): iteration i runs at the closed form's value for i
            loss, _ = segmentation_step(model, criterion, opt, images, labels, input_ids=_batch_prompt(prompt, cache, images.shape[0]),
                                        lr=cosine_lr(args.lr, args.lr_min, iter_num, max_iters))
            if iter_num % 10 == 0:                              # :150-153, without the .item(): device scalars, flushed at validation time
                logged.append((iter_num, loss, cosine_lr(args.lr, args.lr_min, iter_num + 1, max_iters)))
            iter_num += 1
            n_it += 1
            enq += time.perf_counter() - t1
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        epoch_ms.append({"ms": (time.perf_counter() - t0) * 1e3, "updates": n_it, "loader_wait_ms": (train_pf.wait_s - w0) * 1e3, "enqueue_ms": enq * 1e3,
                         "drain_ms": (time.perf_counter() - t2) * 1e3})
        validate = (epoch > 0 and epoch % args.val_every == 0) or (epoch == max_epoch - 1)
        if epoch + 1 < max_epoch and "resident" not in ab:       # the next epoch's first batches load while this one validates
            dm.set_epoch(epoch + 1)
            batches = iter(train_pf)
        if not validate:
            continue

        # Validation (every 10 epochs or last epoch)
        model.eval()
        for it, l, lr in logged:
            writer.add_scalar(f"{args.exp}/train_loss", l, it)
            writer.add_scalar(f"{args.exp}/lr", lr, it)
        last_train = float(logged[-1][1]) if logged else float("nan")
        logged = []
        accumulator = MetricAccumulator(type="seg", criterion=criterion, num_classes=args.num_classes)
        evaluate(model, val_pf, prompt, cache, args, accumulator)
        stats = accumulator.compute()
        accumulator.reset()
        for k in ("loss", "dice_mean", "iou_mean", "hd95_mean", "asd_mean"):
            writer.add_scalar(f"{args.exp}/val_{k.replace('_mean', '')}", stats[k], iter_num)
        writer.flush()

        # Save best model (:190-199)
        stop = False
        if stats["dice_mean"] > best_val_dice:
            patience_counter = 0
            best_val_dice = stats["dice_mean"]
            if rank == 0:
                torch.save({"decoder": model.decoder.state_dict()}, os.path.join(args.train_snapshot_path, "best_model.pth"))
        else:
            patience_counter += 1
        if patience_counter >= args.patience:
            logging.info(f"\nEarly stopping at epoch {epoch + 1}")
            logging.info(f"Early stopping triggered at epoch {epoch + 1}")
            stop = True
        if stop:
            break
        logging.info(f"\titer: {iter_num}, loss: {stats['loss']:.4f}, dice: {stats['dice_mean'] * 100:.2f}, "
                     f"iou: {stats['iou_mean'] * 100:.2f}, hd95: {stats['hd95_mean']:.2f}, asd: {stats['asd_mean']:.2f}")

        # Testing (:212-231)
        evaluate(model, test_pf, prompt, cache, args, accumulator)
        tstats = accumulator.compute()
        accumulator.reset()
        for k in ("loss", "dice_mean", "iou_mean", "hd95_mean", "asd_mean"):
            writer.add_scalar(f"{args.exp}/test_{k.replace('_mean', '')}", tstats[k], iter_num)
        writer.flush()
        model.train()                                            # :234

    writer.close()

    measure = None
    if "post" in ab or "attrib" in ab:                           # measurement knobs (tools/ab_clipseg_post.sh, ab_clipseg_attrib.sh): src/models/clipseg/_measure.py
        from src.models.clipseg import _measure
        measure = _measure.Probe(model, criterion, opt, prompt, cache, args, train_pf)
        measure.before_shutdown(ab)
    for pf in (train_pf, val_pf, test_pf):
        pf.close()                                               # an epoch prefetched and then abandoned by early stopping
    dm.shutdown()
    if measure is not None:
        measure.after_shutdown(ab)
    if world > 1:
        from uia_hip import ops
        import torch.distributed as dist
        dist.barrier()
        ops.comm_destroy()
    if rank == 0 and not os.path.exists(os.path.join(args.train_snapshot_path, "best_model.pth")):
        # no validation ever beat a mean Dice of 0.0: the reference has no checkpoint then and fails in test() on the missing file.  Keep the last iterate instead.
        logging.warning("no validation improved on a mean Dice of 0.0: saving the last iterate as best_model.pth")
        torch.save({"decoder": model.decoder.state_dict()}, os.path.join(args.train_snapshot_path, "best_model.pth"))
    out = {"iters": iter_num, "best_val_dice": best_val_dice, "rank": rank, "world": world, "epochs": epoch_ms}
    if sampler is not None:
        out["power"] = sampler.stop()
    if measure is not None:
        out.update(measure.results)
    if args.stats_json and rank == 0:
        import json
        with open(args.stats_json, "w") as f:
            json.dump(out, f)
    return out


@torch.no_grad()
def test(args):
    logging.info("Start testing")
    rank, _, world = dist_env()
    prompt = get_prompt(args)
    if prompt is None:
        raise ValueError(f"no prompt for --dataset {args.dataset}")
    dm = dataset_seg.DataModule(args, rank=0, world=1)           # every rank evaluates the whole split
    testloader = dm.test_dataloader()
    dm.start_workers()
    bind_device(args)
    UF.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model = prepare_model(args)
    saved_best = os.path.join(args.train_snapshot_path, "best_model.pth")
    adapter_state_dict = torch.load(saved_best, map_location="cpu")
    model.decoder.load_state_dict(adapter_state_dict["decoder"])    # :245-248
    model.eval()
    prompt = prompt.to(args.device)

    if rank == 0:
        fresh_viz_dir(args)                                      # run-directory layout (:256-260)

    accumulator = MetricAccumulator(type="seg", criterion=criterion, num_classes=args.num_classes)
    test_pf = DevicePrefetcher(testloader, None, args.device, second=dataset_seg.second_of)
    evaluate(model, test_pf, prompt, {}, args, accumulator)
    stats = accumulator.compute()
    accumulator.reset()
    test_pf.close()
    dm.shutdown()
    stats["results_csv"] = report_test(args, stats, saved_best, rank)      # :264-306
    return stats


def main(argv=None):
    args = get_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    snapshot_path_list = [f"runs/{args.exp}/{args.dataset}/train", f"runs/{args.exp}/{args.dataset}/test"]
    for path in snapshot_path_list:
        os.makedirs(path, exist_ok=True)
    args.train_snapshot_path, args.test_snapshot_path = snapshot_path_list

    out = {}
    if not args.test:
        setup_logging(args, args.train_snapshot_path)
        out["train"] = train(args)
    setup_logging(args, args.test_snapshot_path)
    out["test"] = test(args)
    return out


if __name__ == "__main__":
    main()
