#!/usr/bin/env python3
"""Known-answer vectors for MONAI's DiceCELoss / compute_dice as the reference uses them — test infrastructure only.

    /root/reference/src/models/clipseg/segmentation.py:84   DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True,
                                                                       smooth_nr=1e-8, smooth_dr=1e-8)
    /root/reference/src/utils/tools.py:185-206, 148-151      compute_dice(one_hot(argmax), label, include_background=False),
                                                             non-finite entries (empty ground truth) dropped

MONAI 1.5.1 (uv.lock:924) is not installed in the build container and is not vendored by the reference, so these vectors
are NOT outputs of MONAI: they restate its published algorithm (monai/losses/dice.py: DiceLoss.forward, DiceCELoss.forward;
monai/metrics/meandice.py: DiceHelper) INDEPENDENTLY of oracle/losses_ref.py — float64 numpy with explicit per-pixel loops,
no torch, no autograd: the gradient is a central finite difference of the float64 loss.  Cases are deliberately asymmetric
(empty ground truth, empty prediction, three classes, H != W) so that an axis mix-up or a wrong reduction cannot cancel.

Run here:  python oracle/gen_dice_golden.py   ->  tests/golden/dicece_cases.npz
"""
import math
import os

import numpy as np


def softmax_pixel(v):
    m = max(v)
    e = [math.exp(x - m) for x in v]
    s = sum(e)
    return [x / s for x in e]


def dicece(logits, label, smooth_nr=1e-8, smooth_dr=1e-8):
    """logits [B][C][H][W] float64, label [B][H][W] int.  Per-pixel loops on purpose."""
    B, C, H, W = logits.shape
    dice_terms = []
    ce_sum, n_pix = 0.0, 0
    for b in range(B):
        inter = [0.0] * C
        pred_o = [0.0] * C
        ground_o = [0.0] * C
        for i in range(H):
            for j in range(W):
                p = softmax_pixel([logits[b, c, i, j] for c in range(C)])
                k = int(label[b, i, j])
                for c in range(C):
                    t = 1.0 if c == k else 0.0
                    inter[c] += p[c] * t
                    pred_o[c] += p[c] * p[c]                  # squared_pred=True
                    ground_o[c] += t * t
                ce_sum += -math.log(p[k])                       # CrossEntropyLoss(reduction="mean") over all pixels of the batch
                n_pix += 1
        for c in range(C):                                      # include_background=True: every channel counts
            dice_terms.append(1.0 - (2.0 * inter[c] + smooth_nr) / (pred_o[c] + ground_o[c] + smooth_dr))
    return sum(dice_terms) / len(dice_terms) + ce_sum / n_pix   # reduction="mean" over (B, C); lambda_dice = lambda_ce = 1


def fd_grad(logits, label, h=1e-6):
    g = np.zeros_like(logits)
    it = np.nditer(logits, flags=["multi_index"])
    for _ in it:
        idx = it.multi_index
        old = logits[idx]
        logits[idx] = old + h
        up = dicece(logits, label)
        logits[idx] = old - h
        dn = dicece(logits, label)
        logits[idx] = old
        g[idx] = (up - dn) / (2 * h)
    return g


def dice_metric(logits, label):
    """Per-image Dice of class 1 for the argmax mask (include_background=False; binary problems of the reference); NaN when the
    ground truth holds no foreground pixel (ignore_empty=True)."""
    B, C, H, W = logits.shape
    out = []
    for b in range(B):
        npred = ngt = nboth = 0
        for i in range(H):
            for j in range(W):
                k = max(range(C), key=lambda c: (logits[b, c, i, j], -c))        # argmax, first index on ties (torch.argmax)
                pr, gt = (k == 1), (int(label[b, i, j]) == 1)
                npred += pr
                ngt += gt
                nboth += pr and gt
        out.append(float("nan") if ngt == 0 else 2.0 * nboth / (ngt + npred))
    return np.array(out)


def main():
    rng = np.random.default_rng(20251003)
    cases = {}
    # A: binary, image 0 has an EMPTY ground truth, image 1 a small blob; H != W
    la = rng.normal(0, 1.5, (2, 2, 5, 7))
    ya = np.zeros((2, 5, 7), dtype=np.int64)
    ya[1, 1:3, 2:6] = 1
    cases["A_empty_gt"] = (la, ya)
    # B: binary, the PREDICTION is empty (class 0 wins everywhere by a wide margin) while the ground truth is not
    lb = rng.normal(0, 0.3, (2, 2, 4, 6))
    lb[:, 0] += 6.0
    yb = np.zeros((2, 4, 6), dtype=np.int64)
    yb[0, :2, :3] = 1
    yb[1, 3, 1:5] = 1
    cases["B_empty_pred"] = (lb, yb)
    # C: three classes, unbalanced labels, one class absent from image 1
    lc = rng.normal(0, 2.0, (2, 3, 3, 4))
    yc = rng.integers(0, 3, (2, 3, 4))
    yc[1][yc[1] == 2] = 0
    cases["C_three_class"] = (lc, yc)
    # D: binary, single image, every pixel foreground (empty background channel)
    ld = rng.normal(0, 1.0, (1, 2, 3, 3))
    yd = np.ones((1, 3, 3), dtype=np.int64)
    cases["D_all_foreground"] = (ld, yd)
    out = {}
    for name, (logits, label) in cases.items():
        logits = np.ascontiguousarray(logits, dtype=np.float64)
        out[name + "_logits"] = logits.astype(np.float32)
        out[name + "_label"] = label.astype(np.int64)
        l32 = out[name + "_logits"].astype(np.float64)                         # the vectors are quoted for the float32-rounded inputs
        out[name + "_loss"] = np.array(dicece(l32, label))
        out[name + "_grad"] = fd_grad(l32.copy(), label)
        if logits.shape[1] == 2:
            out[name + "_dice"] = dice_metric(l32, label)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dicece_cases.npz")
    np.savez_compressed(path, **out)
    for name in cases:
        print(name, "loss", float(out[name + "_loss"]), "dice", out.get(name + "_dice"))


if __name__ == "__main__":
    main()
