"""DiceCE loss and Dice metric with MONAI 1.5.1 semantics (reference: /root/reference/src/models/clipseg/segmentation.py:84,
`DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=1e-8, smooth_dr=1e-8)`; metric src/utils/tools.py:185-206).
SURVEY §8(f)-3 lists an on-device fused kernel as a follow-up; for now these are plain device-side torch ops at the edge of the
hot path (the decoder gradient enters libuia_hip.so through `logits.grad`)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DiceCELoss(nn.Module):
    def __init__(self, smooth_nr=1e-8, smooth_dr=1e-8):
        super().__init__()
        self.smooth_nr, self.smooth_dr = smooth_nr, smooth_dr

    def forward(self, logits, label):
        """logits [B, C, H, W]; label [B, 1, H, W] with class indices."""
        p = torch.softmax(logits, dim=1)
        t = F.one_hot(label[:, 0].long(), logits.shape[1]).permute(0, 3, 1, 2).to(p.dtype)
        inter = (p * t).sum(dim=(2, 3))
        den = (p * p).sum(dim=(2, 3)) + (t * t).sum(dim=(2, 3))
        dice = 1.0 - (2.0 * inter + self.smooth_nr) / (den + self.smooth_dr)
        return dice.mean() + F.cross_entropy(logits, label[:, 0].long())


def dice_per_image(logits, label):
    """compute_dice(one_hot(argmax), label, include_background=False): NaN where the ground truth is empty."""
    pred = logits.argmax(dim=1) == 1
    gt = label[:, 0] > 0
    inter = (pred & gt).flatten(1).sum(1).double()
    tot = pred.flatten(1).sum(1).double() + gt.flatten(1).sum(1).double()
    out = 2 * inter / tot
    out[gt.flatten(1).sum(1) == 0] = float("nan")
    return out
