"""DiceCE loss and Dice metric with MONAI 1.5.1 semantics (reference: /root/reference/src/models/clipseg/segmentation.py:84,
`DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=1e-8, smooth_dr=1e-8)`; metric src/utils/tools.py:185-206).
The loss runs fused on the device (`uia_dicece_fwd_bwd`: loss and d loss / d logits in one call); the Dice metric
is a handful of torch reductions on the validation path."""
import torch
import torch.nn as nn

from uia_hip import ops


class _DiceCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, nr, dr):
        loss, dl = ops.dicece_fwd_bwd(logits.contiguous().float(), label, nr, dr)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None


class DiceCELoss(nn.Module):
    def __init__(self, smooth_nr=1e-8, smooth_dr=1e-8):
        super().__init__()
        self.smooth_nr, self.smooth_dr = smooth_nr, smooth_dr

    def forward(self, logits, label):
        """logits [B, C, H, W]; label [B, 1, H, W] with class indices."""
        return _DiceCEFn.apply(logits, label, self.smooth_nr, self.smooth_dr)


def dice_per_image(logits, label):
    """compute_dice(one_hot(argmax), label, include_background=False): NaN where the ground truth is empty."""
    pred = logits.argmax(dim=1) == 1
    gt = label[:, 0] > 0
    inter = (pred & gt).flatten(1).sum(1).double()
    tot = pred.flatten(1).sum(1).double() + gt.flatten(1).sum(1).double()
    out = 2 * inter / tot
    out[gt.flatten(1).sum(1) == 0] = float("nan")
    return out


def iou_per_image(logits, label):
    """compute_iou(one_hot(argmax), label, include_background=False, ignore_empty=True): n(P∩G) / n(P∪G) per image, NaN where the ground truth is empty
    (reference src/utils/tools.py:17, :192)."""
    pred = logits.argmax(dim=1) == 1
    gt = label[:, 0] > 0
    inter = (pred & gt).flatten(1).sum(1).double()
    union = pred.flatten(1).sum(1).double() + gt.flatten(1).sum(1).double() - inter
    out = inter / union
    out[gt.flatten(1).sum(1) == 0] = float("nan")
    return out
