"""Contrastive loss on the HIP path (drop-in for /root/reference/src/losses/losses.py).

``InfoNCELoss(temperature=0.07)(image_features, text_features, batch_size=None)`` returns the scalar
symmetric InfoNCE; forward and the gradient w.r.t. both feature matrices come from one C-ABI call
(``uia_infonce_fwd_bwd``), so the backward is a scale by the incoming gradient.
"""
import torch
import torch.nn as nn

from uia_hip import ops


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, temperature):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss, dimg, dtxt = ops.infonce(img.contiguous().float(), txt.contiguous().float(), temperature, want_grads=need)
        if need:
            ctx.save_for_backward(dimg, dtxt)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dimg, dtxt = ctx.saved_tensors
        return (dimg * g if ctx.needs_input_grad[0] else None), (dtxt * g if ctx.needs_input_grad[1] else None), None


class InfoNCELoss(nn.Module):
    """Standard InfoNCE / NT-Xent (reference :10-47): L2-normalise, logits = I·Tᵀ/τ, CE both ways, mean."""

    def __init__(self, temperature=0.07):
        super().__init__()
        self.temperature = temperature

    def forward(self, image_features, text_features, batch_size=None):
        if batch_size is not None and batch_size != image_features.shape[0]:
            raise ValueError("batch_size must equal the number of feature rows (reference losses.py:37 builds arange(batch_size) labels)")
        return _InfoNCEFn.apply(image_features, text_features, self.temperature)
