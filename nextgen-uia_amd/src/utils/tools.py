"""Logging / model summary / metric helpers used by the entry points (subset of /root/reference/src/utils/tools.py:37-206;
HD95 / ASD (MONAI + scipy distance transforms on the host) and visualisation stay out of the hot path)."""
import ast
import logging
import os
import sys


def setup_logging(args, snapshot_path):
    """Log to <snapshot_path>/log.log and stdout (reference :37-52)."""
    os.makedirs(snapshot_path, exist_ok=True)
    root = logging.getLogger()
    for h in list(root.handlers):
        root.removeHandler(h)
    logging.basicConfig(filename=os.path.join(snapshot_path, "log.log"), level=logging.INFO,
                        format="[%(asctime)s.%(msecs)03d] %(message)s", datefmt="%H:%M:%S")
    root.addHandler(logging.StreamHandler(sys.stdout))
    logging.info(str(args))


def model_summary(models):
    """Parameter counts per named model (reference :69-105)."""
    lines = []
    for name, model in models.items():
        total = sum(p.numel() for p in model.parameters())
        train = sum(p.numel() for p in model.parameters() if p.requires_grad)
        lines.append(f"{name}: total {total:,} | trainable {train:,} | frozen {total - train:,}")
    return "\n".join(lines)


def parse_config(text):
    """Safe parser for the entry points' --model_config: Python literals plus `dict(key=value, ...)` calls (nothing else is
    evaluated), e.g. "dict(embed_dim=128, vision_cfg=dict(img_size=32, patch_size=8))"."""
    def ev(node):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "dict" and not node.args:
            if any(kw.arg is None for kw in node.keywords):
                raise ValueError("** expansion is not accepted in a model config")
            return {kw.arg: ev(kw.value) for kw in node.keywords}
        if isinstance(node, ast.Dict):
            return {ev(k): ev(v) for k, v in zip(node.keys, node.values)}
        if isinstance(node, (ast.List, ast.Tuple)):
            vals = [ev(e) for e in node.elts]
            return vals if isinstance(node, ast.List) else tuple(vals)
        return ast.literal_eval(node)
    return ev(ast.parse(text.strip(), mode="eval").body)


def default_device():
    """The reference's `--device` default ("cuda:0" if a GPU is there, tools.py:24) decided WITHOUT a HIP call: the loader workers must fork before this
    process owns a GPU context, and on ROCm builds torch.cuda.is_available() / device_count() may create one."""
    hidden = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", None))
    return "cuda:0" if os.path.exists("/dev/kfd") and hidden != "" else "cpu"


class ScalarLog:
    """Stands where the reference's TensorBoard SummaryWriter stands (tensorboard is not part of the build image): add_scalar appends one JSON line to
    <dir>/scalars.jsonl; values may be device scalars — they are converted only when flush() is called, so logging never synchronises the loop."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path, self.rows = os.path.join(log_dir, "scalars.jsonl"), []

    def add_scalar(self, tag, value, step):
        self.rows.append((tag, value, int(step)))

    def add_images(self, *a, **k):                              # visualisation: out of scope
        pass

    def flush(self):
        import json
        if self.rows:
            with open(self.path, "a") as f:
                for tag, v, step in self.rows:
                    f.write(json.dumps({"tag": tag, "value": float(v), "step": step}) + "\n")
            self.rows = []

    def close(self):
        self.flush()


class MetricAccumulator:
    """MetricAccumulator(type="seg") of the reference (tools.py:108-206) with the per-batch work on the device: update() enqueues the criterion, the
    per-image Dice and IoU of the arg-max mask (losses/dice.py, MONAI semantics) and keeps the results as device tensors; compute() is the one host read and
    returns the reference's keys — means / stds over the FINITE per-image values (np.std, population form, :148-163), `loss` the mean of the finite per-batch
    losses.  hd95_* / asd_* are NaN: MONAI's surface distances are host-side scipy work outside this build."""

    def __init__(self, type="seg", criterion=None, num_classes=2):
        if type != "seg":
            raise NotImplementedError("only the segmentation accumulator is part of this build (cls / recon metrics are torchmetrics / MONAI host code)")
        self.type, self.criterion, self.num_classes = type, criterion, num_classes
        self.reset()

    def reset(self):
        self._dice, self._iou, self._loss = [], [], []

    def update(self, preds, labels):
        from src.losses.dice import dice_per_image, iou_per_image
        self._loss.append(self.criterion(preds.float(), labels.float()).detach().double().reshape(1))
        self._dice.append(dice_per_image(preds, labels))
        self._iou.append(iou_per_image(preds, labels))

    def compute(self):
        import numpy as np
        import torch
        if not self._dice:
            nan = float("nan")
            return {k: nan for k in ("dice_mean", "dice_std", "iou_mean", "iou_std", "hd95_mean", "hd95_std", "asd_mean", "asd_std", "loss")}
        n = sum(t.numel() for t in self._dice)
        flat = torch.cat(self._dice + self._iou + self._loss).cpu().numpy()       # one device-to-host copy
        dice, iou, loss = flat[:n], flat[n:2 * n], flat[2 * n:]
        fin = lambda a: a[np.isfinite(a)]
        nan = float("nan")
        return {"dice_mean": float(np.mean(fin(dice))), "dice_std": float(np.std(fin(dice))), "iou_mean": float(np.mean(fin(iou))), "iou_std": float(np.std(fin(iou))),
                "hd95_mean": nan, "hd95_std": nan, "asd_mean": nan, "asd_std": nan, "loss": float(np.mean(fin(loss)))}


def report_test(args, stats, saved_best, rank=0):
    """The tail of the segmentation entry points' test() (reference src/models/clipseg/segmentation.py:275-306, biomedclip/segmentation.py:322-353): the
    Metric / Mean / Std table (Dice and IoU in percent) to the log, then runs/<exp>/<dataset>/test/<time>_iou=<iou>/ with results.csv
    (DataFrame.to_csv(index=False, float_format="%.2f"): NaN as an empty field), a copy of the checkpoint, the viz folder and the log."""
    import datetime
    import shutil
    rows = [("Dice", stats["dice_mean"] * 100, stats["dice_std"] * 100), ("IoU", stats["iou_mean"] * 100, stats["iou_std"] * 100),
            ("HD95", stats["hd95_mean"], stats["hd95_std"]), ("ASD", stats["asd_mean"], stats["asd_std"])]
    table = f"{'Metric':>6} {'Mean':>6} {'Std':>6}\n" + "".join(f"{m:>6} {a:6.2f} {s_:6.2f}\n" for m, a, s_ in rows)
    logging.info(f"\n{'=' * 50}\n" + table + f"{'=' * 50}\n")
    if rank != 0:
        return None
    backup_folder = os.path.join(args.test_snapshot_path, f"{datetime.datetime.now().strftime('%Y_%m_%d_%H_%M_%S')}_iou={stats['iou_mean'] * 100:.2f}")
    base, n = backup_folder, 1
    while os.path.exists(backup_folder):                        # two tests within one second with the same IoU (the reference's os.makedirs raises there)
        n += 1
        backup_folder = f"{base}__{n}"
    os.makedirs(backup_folder)
    csv_path = os.path.join(backup_folder, "results.csv")
    with open(csv_path, "w") as f:
        f.write("Metric,Mean,Std\n")
        for m, a, s_ in rows:
            f.write(f"{m},{'' if a != a else '%.2f' % a},{'' if s_ != s_ else '%.2f' % s_}\n")
    logging.info(f"Results saved to: {csv_path}")
    shutil.copy(saved_best, os.path.join(backup_folder, "best_model.pth"))
    viz_path = args.test_snapshot_path + "/viz"
    if os.path.exists(viz_path):
        shutil.move(viz_path, os.path.join(backup_folder, "viz"))
    for h in list(logging.getLogger().handlers):
        h.flush()
    shutil.move(os.path.join(args.test_snapshot_path, "log.log"), os.path.join(backup_folder, "log.log"))
    return csv_path


def fresh_viz_dir(args):
    """runs/<exp>/<dataset>/test/viz, emptied (reference clipseg/segmentation.py:256-260); visualize_seg's matplotlib overlays themselves are outside this build."""
    import shutil
    viz_path = args.test_snapshot_path + "/viz"
    if os.path.exists(viz_path):
        shutil.rmtree(viz_path)
    os.makedirs(viz_path)
    return viz_path
