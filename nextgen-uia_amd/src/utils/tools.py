"""Logging / model summary helpers used by the entry points (subset of /root/reference/src/utils/tools.py:37-105;
metrics and visualisation stay out of the hot path)."""
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
