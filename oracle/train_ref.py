"""One optimiser update of the contrastive fine-tune loop — CPU restatement.  Test infrastructure only.

Follows /root/reference/src/models/biomedclip/finetune.py:272-302:
  features -> InfoNCE (:276-279) -> (loss / accumulation_steps).backward() (:287-288), summed over
  the micro-batches of one cycle; clip_grad_norm_(max_norm) over the trainable grads (:297-298);
  AdamW(lr, betas, weight_decay) step (:244-249, :300).
Data-parallel equivalence: world ranks x 1 micro-batch == 1 process x world
micro-batches, which is what `micro_batches` expresses.
"""
import math
import torch

from . import losses_ref, vit_ref, text_ref


def biomedclip_loss(P, images, ids, mona=None, lora=None, temperature=0.07, heads=12, text_heads=12):
    img = vit_ref.timm_vit_forward(images, P, heads=heads, mona=mona, lora=lora)
    txt = text_ref.bert_text_forward(ids, P, heads=text_heads, lora=lora)      # LoRA factors in the text tower are used where present
    return losses_ref.info_nce(img, txt, temperature)


def grads_of(loss_fn, P, trainable, micro_batches):
    """Accumulated grads over micro-batches of mean-loss / len(micro_batches) (finetune.py:287-288)."""
    leaves = {k: P[k].detach().clone().requires_grad_(True) for k in trainable}
    Pq = dict(P)
    Pq.update(leaves)
    total = 0.0
    for mb in micro_batches:
        loss = loss_fn(Pq, *mb)
        (loss / len(micro_batches)).backward()
        total += float(loss)
    return {k: v.grad for k, v in leaves.items()}, total / len(micro_batches)


def clip_and_adamw(params, grads, m, v, step, lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (coef = max_norm/(norm+1e-6), clamped to 1) then torch AdamW.
    All dicts are keyed alike; updated in place.  `step` is 1-based."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    coef = min(1.0, max_norm / (total + 1e-6)) if max_norm > 0 else 1.0
    b1, b2 = betas
    for k in params:
        g = grads[k] * coef
        params[k].mul_(1 - lr * weight_decay)
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)
    return total
