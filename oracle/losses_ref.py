"""Contrastive / segmentation losses — functional CPU restatement.  Test infrastructure only.

InfoNCE follows /root/reference/src/losses/losses.py:23-47.
DiceCE / Dice follow MONAI 1.5.1's published definitions as used at
/root/reference/src/models/clipseg/segmentation.py:84 and src/utils/tools.py:185-206
(MONAI is absent from the container: PARITY UNPINNED, hand-computed cases only).
"""
import torch
import torch.nn.functional as F


def info_nce(img, txt, temperature=0.07):
    i = img / img.norm(dim=1, keepdim=True).clamp_min(1e-12)          # F.normalize eps, :25-26
    t = txt / txt.norm(dim=1, keepdim=True).clamp_min(1e-12)
    logits = i @ t.T / temperature                                     # :34
    lse_r = torch.logsumexp(logits, dim=1)
    lse_c = torch.logsumexp(logits, dim=0)
    diag = logits.diagonal()
    return 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())       # :41-45


def dice_ce(logits, label, smooth_nr=1e-8, smooth_dr=1e-8):
    """DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True); logits [B,2,H,W], label [B,1,H,W]."""
    p = torch.softmax(logits, dim=1)
    t = F.one_hot(label[:, 0].long(), logits.shape[1]).permute(0, 3, 1, 2).to(p.dtype)
    inter = (p * t).sum(dim=(2, 3))
    den = (p * p).sum(dim=(2, 3)) + (t * t).sum(dim=(2, 3))
    dice = 1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)
    ce = F.cross_entropy(logits, label[:, 0].long())
    return dice.mean() + ce


def dice_metric(logits, label):
    """Per-image Dice of the argmax mask for class 1; NaN when the ground truth is empty."""
    pred = logits.argmax(dim=1) == 1
    gt = label[:, 0] > 0
    inter = (pred & gt).flatten(1).sum(1).double()
    tot = pred.flatten(1).sum(1).double() + gt.flatten(1).sum(1).double()
    out = 2 * inter / tot
    out[gt.flatten(1).sum(1) == 0] = float("nan")
    return out
