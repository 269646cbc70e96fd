"""Logging / model summary helpers used by the entry points (subset of /root/reference/src/utils/tools.py:37-105;
metrics and visualisation stay out of the hot path)."""
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
