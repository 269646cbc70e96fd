"""Training-step plumbing for the adapter fine-tune loop: a FLAT fp32 buffer holding every trainable
(adapter) parameter, its gradient twin, one fused clip+AdamW kernel over it, and — data parallel — one
RCCL all-reduce of the gradient buffer per optimiser step.

Reference semantics reproduced (/root/reference/src/models/biomedclip/finetune.py):
  :244-249  AdamW(trainable params, lr, betas=(0.9,0.95), weight_decay)      :255  CosineAnnealingLR(T_max, eta_min)
  :287-288  (loss / accumulation_steps).backward(), gradients SUMMED over the micro-batches of a cycle
  :297-302  clip_grad_norm_(max_norm) → optimizer.step() → scheduler.step() → zero_grad()
Data parallel over R ranks ≡ the reference with accumulation_steps = R: every rank back-propagates
its local mean loss, the flat gradient buffers are summed by one all-reduce, and the update uses grad/R.
"""
import math
import os

import torch

from . import functional as UF
from . import ops


class FlatAdapterOptimizer:
    def __init__(self, named_params, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, max_norm=1.0):
        self.names = [k for k, _ in named_params]
        self.params = [p for _, p in named_params]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        assert dev.type == "cuda", "the fused optimiser runs on the GPU only"
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4                     # keep every view 16-byte aligned
        self.numel = n
        self.p = torch.zeros(n, device=dev)
        self.g = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.ws = torch.zeros(2, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.p[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g[off:off + p.numel()].view(p.shape)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.steps = 0
        self.world = 1

    def zero_grad(self):
        self.g.zero_()

    def all_reduce(self):
        if self.world > 1:
            ops.allreduce_sum(self.g)

    def step(self, lr=None, grad_scale=None):
        """One update from the accumulated gradient buffer (already all-reduced when world > 1)."""
        self.steps += 1
        gs = (1.0 / self.world) if grad_scale is None else grad_scale
        ops.adamw_clip_step(self.p, self.g, self.m, self.v, self.lr if lr is None else lr, self.betas, self.eps, self.weight_decay,
                            self.max_norm, self.steps, gs, self.ws)
        UF.WEIGHTS.bump()                                      # T copies of the adapter weights are stale now

    def grad_norm(self):
        return math.sqrt(float(self.ws[0]))

    def state_dict_named(self):
        return {k: p.detach().clone() for k, p in zip(self.names, self.params)}


def cosine_lr(base_lr, eta_min, t, t_max):
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (finetune.py:255)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2


# ------------------------------------------------------------------------------------------------ data parallel
def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_data_parallel(opt=None):
    """One process per GPU.  torch.distributed (RCCL backend) is used for rendezvous, barriers and the unique-id
    broadcast only; the gradient all-reduce itself is the library's own RCCL call on the compute stream."""
    import torch.distributed as dist
    rank, local, world = dist_env()
    torch.cuda.set_device(local)
    if world > 1:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        uid = [ops.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ops.comm_init(rank, world, uid[0])
    if opt is not None:
        opt.world = world
        if world > 1:                                          # identical adapter weights on every rank
            import torch.distributed as dist
            dist.broadcast(opt.p, src=0)
    return rank, local, world


class GatherFeaturesFn(torch.autograd.Function):
    """[B, E] local features → [world·B, E] in rank order (RCCL all-gather on the compute stream).  Every rank then evaluates the
    SAME global-batch loss, so the gradient of that loss with respect to this rank's rows is simply the local slice of the
    gathered gradient — no collective in the backward; the parameter gradients of the ranks add up in the usual all-reduce."""

    @staticmethod
    def forward(ctx, x, rank, world):
        x = x.contiguous()
        out = torch.empty(world * x.shape[0], x.shape[1], device=x.device, dtype=x.dtype)
        if world > 1:
            ops.allgather(x, out)
        else:
            out.copy_(x)
        ctx.meta = (rank, x.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        rank, B = ctx.meta
        return g[rank * B:(rank + 1) * B].contiguous(), None, None


_SIDE_STREAM = {}


def _side_stream(device):
    if device not in _SIDE_STREAM:
        _SIDE_STREAM[device] = torch.cuda.Stream(device=device)
    return _SIDE_STREAM[device]


def contrastive_step(model, criterion, opt, images, ids, micro_batches=1, lr=None, overlap_text=True, global_loss=False):
    """One optimiser update: encode → InfoNCE → backward (→ all-reduce) → clip+AdamW.  Returns the loss tensor (device).

    global_loss (opt-in, not the reference's semantics): the InfoNCE batch is the GLOBAL batch — features are all-gathered,
    every rank evaluates the same world·B × world·B loss, back-propagates the rows it owns, and the all-reduced parameter
    gradient is the gradient of that one loss (no 1/world scaling).  Default: each rank's local loss, averaged (≡ the
    reference's gradient accumulation).

    overlap_text: the frozen text tower does not depend on the image tower, so it runs on a second HIP stream beside
    encode_image (same kernels, same results); the streams join before the loss."""
    UF.clear_t_copies()
    opt.zero_grad()
    total = None
    mb = images.shape[0] // micro_batches
    cur = torch.cuda.current_stream()
    for i in range(micro_batches):
        im, tk = images[i * mb:(i + 1) * mb], ids[i * mb:(i + 1) * mb]
        if overlap_text:
            side = _side_stream(images.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ft = model.encode_text(tk)
            fi = model.encode_image(im)
            cur.wait_stream(side)
            ft.record_stream(cur)
        else:
            fi = model.encode_image(im)
            ft = model.encode_text(tk)
        if global_loss:
            rank, _, _ = dist_env()
            fi, ft = GatherFeaturesFn.apply(fi, rank, opt.world), GatherFeaturesFn.apply(ft, rank, opt.world)
        loss = criterion(fi, ft)
        (loss / micro_batches).backward()
        total = loss.detach() if total is None else total + loss.detach()
    opt.all_reduce()
    opt.step(lr=lr, grad_scale=1.0 if global_loss else None)
    return total / micro_batches
