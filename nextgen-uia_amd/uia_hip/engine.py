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
import time

import torch

from . import functional as UF
from . import ops


class FlatLayout:
    """Where every trainable parameter lives inside the flat fp32 buffers: element offsets (each view 16-byte aligned) and the
    four buffers themselves.  Device-agnostic on purpose — the world-size-2 gloo test (tests/test_dp_gloo.py) drives exactly
    this code on CPU tensors; the optimiser below adds the fused HIP update on top of it."""

    ALIGN = 4                                                   # fp32 elements per 16 bytes

    def __init__(self, named_params):
        self.names = [k for k, _ in named_params]
        self.params = [p for _, p in named_params]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        assert all(p.device == dev for p in self.params), "trainable parameters must live on ONE device (one process per GPU)"
        assert all(p.dtype == torch.float32 for p in self.params), "adapter parameters are fp32 masters"
        self.device = dev
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = n
        self.p = torch.zeros(n, device=dev)
        self.g = torch.zeros(n, device=dev)
        self.acc = torch.zeros(n + self.ALIGN, device=dev)     # guarded loops: the update cycle's accumulator; acc[n] = non-finite flag of the boundary micro-batch (travels through the all-reduce)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.p[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g[off:off + p.numel()].view(p.shape)

    def grad_views_intact(self):
        """True while every parameter's .grad is still the view of the flat gradient buffer handed out at construction
        (an `optimizer.zero_grad(set_to_none=True)` or a re-assigned .grad breaks it)."""
        g0 = self.g.data_ptr()
        return all(p.grad is not None and p.grad.data_ptr() == g0 + 4 * off for p, off in zip(self.params, self.offsets))

    def rebind_grads(self):
        for p, off in zip(self.params, self.offsets):
            p.grad = self.g[off:off + p.numel()].view(p.shape)

    def unflatten(self, flat):
        return {k: flat[off:off + p.numel()].view(p.shape) for k, p, off in zip(self.names, self.params, self.offsets)}


def dp_grad_scale(world, global_loss=False):
    """Factor applied to the all-reduced (SUMMED) gradient buffer: 1/world for the reference-equivalent local loss (every
    rank back-propagates its own mean loss ≡ finetune.py:287-288 with accumulation_steps = world); 1 for the opt-in
    global-batch loss, whose per-rank contributions already add up to the gradient of the one global loss."""
    return 1.0 if global_loss else 1.0 / world


class FlatAdapterOptimizer(FlatLayout):
    def __init__(self, named_params, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, max_norm=1.0):
        super().__init__(named_params)
        assert self.device.type == "cuda", "the fused optimiser runs on the GPU only"
        self.ws = torch.zeros(2, device=self.device)
        # guarded loops (accumulate / update below): [updates applied, micro-batches accumulated, micro-batches skipped, updates skipped], loss sum, scalars of the update
        self.ctl = torch.zeros(4, device=self.device, dtype=torch.int32)
        self.stats = torch.zeros(4, device=self.device)
        self.ws8 = torch.zeros(8, device=self.device)
        self.ok_log = None
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.steps = 0
        self.world = 1
        self.collective = False                                 # True: run the RCCL all-reduce even at world 1 (tests of the comm path)
        for p in self.params:                                   # MonaFn may accumulate straight into these views (functional.mona_apply)
            p._uia_flat_grad = True

    def zero_grad(self, set_to_none=False):
        """Zeroes the flat gradient buffer; the .grad views stay bound (set_to_none is accepted for torch.optim call sites and
        ignored: dropping the views would orphan the buffer the all-reduce and the fused update work on)."""
        self.g.zero_()
        if not self.grad_views_intact():
            self.rebind_grads()

    def _adopt_grads(self):
        """A caller that dropped or replaced the .grad views (zero_grad(set_to_none=True), manual assignment) makes autograd
        write gradients into tensors of its own: fold those into the flat buffer and re-bind the views, so that the
        all-reduce and the fused update never run on a buffer the backward did not fill."""
        if self.grad_views_intact():
            return
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.g[off:off + p.numel()].view(p.shape)
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    view.add_(p.grad.to(view.dtype))
                p.grad = view

    def all_reduce(self):
        UF.join_side_streams()                                 # weight-gradient launches on the side stream (UF.set_wgrad_side_stream) write into self.g
        self._adopt_grads()
        if self.world > 1 or self.collective:
            ops.allreduce_sum(self.g)

    def step(self, lr=None, grad_scale=None):
        """One update from the accumulated gradient buffer (already all-reduced when world > 1)."""
        UF.join_side_streams()
        self._adopt_grads()
        self.steps += 1
        self._last_guarded = False
        gs = dp_grad_scale(self.world) if grad_scale is None else grad_scale
        ops.adamw_clip_step(self.p, self.g, self.m, self.v, self.lr if lr is None else lr, self.betas, self.eps, self.weight_decay,
                            self.max_norm, self.steps, gs, self.ws)
        UF.WEIGHTS.bump()                                      # T copies of the adapter weights are stale now
        if UF.ln_fold_enabled(UF.compute_dtype()):             # guard of the folded LayerNorms: every folded layer of every step was checked on
            UF.poll_ln_flag(self.device)                       # the device; this reads the verdict without a host sync (UF.POLL_LAG steps late)

    def grad_norm(self):
        """‖g‖₂ of the last update (before clipping; one host read)."""
        return math.sqrt(float(self.ws8[0] if getattr(self, "_last_guarded", False) else self.ws[0]))

    # ---- guarded API (round 5): the loop never reads the loss on the host.  One micro-batch: backward into the .grad views (self.g, the staging
    # buffer) -> accumulate(loss) adds it to self.acc when the loss is finite (device decision), zeroes self.g, leaves the flag in acc[n];
    # at an update boundary: update() = all-reduce of self.acc (flag included) + the device-guarded clip + AdamW, which also zeroes acc.
    def start_log(self, n):
        """An epoch's per-micro-batch finite flags (uint8, 2 = not run) for the end-of-epoch log; read with read_guard()."""
        self.ok_log = torch.full((max(1, n),), 2, device=self.device, dtype=torch.uint8)

    def accumulate(self, loss, log_index=None):
        UF.join_side_streams()                                 # weight-gradient launches on the side stream write into self.g
        self._adopt_grads()
        ops.grad_accum_guarded(self.acc, self.g, loss.reshape(1), self.stats, self.ctl, self.ok_log if log_index is not None else None, log_index or 0)
        self._staging_dirty = False                            # the kernel zeroed the staging buffer

    def update(self, lr=None, lr_min=0.0, t_max=0, grad_scale=None, discard_on_skip=False):
        """t_max > 0: cosine schedule from lr down to lr_min over t_max updates, indexed by the DEVICE's count of applied updates.
        discard_on_skip: a skipped update ZEROES the accumulator (loops whose reference calls zero_grad() before every backward — metaclip/finetune.py:166,
        clip/finetune.py — so that under data parallelism the finite ranks' gradients of a skipped iteration are not carried into the next update).
        Default: the accumulator keeps the cycle's sum (biomedclip/finetune.py has no zero_grad without a step)."""
        if self.world > 1 or self.collective:
            ops.allreduce_sum(self.acc)
        self.steps += 1                                        # optimistic host count (the device's ctl[0] is the truth: read_guard())
        self._last_guarded = True
        if getattr(self, "snapshot_grads", False):             # tests / tools: the accumulated (all-reduced, unscaled) gradient of this update, which the update zeroes
            self.last_g = self.acc[:self.numel].clone()
        gs = dp_grad_scale(self.world) if grad_scale is None else grad_scale
        ops.adamw_clip_step_guarded(self.p, self.acc, self.m, self.v, self.lr if lr is None else lr, lr_min, t_max, self.betas, self.eps, self.weight_decay,
                                    self.max_norm, gs, 0.0 if discard_on_skip else 1.0 / self.world, self.ws8, self.ctl)
        UF.WEIGHTS.bump()
        if UF.ln_fold_enabled(UF.compute_dtype()):
            UF.poll_ln_flag(self.device)
        self._poll_norm()

    def _poll_norm(self):
        """The squared gradient norm of every guarded update, read WITHOUT a host sync (copied to pinned memory behind the update, examined one update
        late).  A non-finite norm behind finite losses means the backward itself diverged — or a three-byte gradient token (functional.publish_grad3)
        was read by something that is not its consumer: raise instead of training on NaN weights."""
        pending = self.__dict__.setdefault("_norm_polls", [])
        while len(pending) >= UF.POLL_LAG:                     # recorded UF.POLL_LAG updates ago: returns at once unless the host is that far ahead of the GPU
            prev = pending.pop(0)
            t0 = time.perf_counter()
            prev[1].synchronize()
            self.gpu_wait_s = getattr(self, "gpu_wait_s", 0.0) + time.perf_counter() - t0
            if float(prev[0][1]) != 0.0 and not math.isfinite(float(prev[0][0])):
                raise FloatingPointError("uia_hip: the accumulated adapter gradient is non-finite although every accumulated loss was finite: the backward diverged "
                                         "(or a three-byte gradient token reached a consumer that is not its partner Function — run with engine.GRAD_RESID3 = False)")
        host = torch.empty(2, dtype=torch.float32, pin_memory=True)
        host.copy_(self.ws8[:2], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((host, ev))

    def read_guard(self):
        """One host sync: {updates, accumulated, skipped, updates_skipped, loss_sum, ok_log}; clears the loss sum and the log."""
        c = self.ctl.tolist()
        out = {"updates": c[0], "accumulated": c[1], "skipped": c[2], "updates_skipped": c[3], "loss_sum": float(self.stats[0]),
               "ok_log": None if self.ok_log is None else self.ok_log.tolist()}
        self.stats.zero_()
        self.ok_log = None
        return out

    def state_dict_named(self):
        return {k: p.detach().clone() for k, p in zip(self.names, self.params)}


def cosine_lr(base_lr, eta_min, t, t_max):
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (finetune.py:255)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2


# ------------------------------------------------------------------------------------------------ data parallel
def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def bind_device(args=None):
    """FIRST call of an entry point under torch.distributed.run: binds this process to cuda:LOCAL_RANK *before* the model, the
    flat optimiser buffers or any stream exist, and points `args.device` there (the reference's single `--device` flag,
    finetune.py:101, cannot name a per-rank GPU).  Single process: leaves everything alone."""
    rank, local, world = dist_env()
    if world > 1:
        torch.cuda.set_device(local)
        if args is not None:
            args.device = f"cuda:{local}"
    return rank, local, world


def init_data_parallel(opt=None, force_comm=False):
    """One process per GPU.  torch.distributed (RCCL backend) is used for rendezvous, barriers, the unique-id broadcast and the
    scalar control-flow agreements below; the gradient all-reduce itself is the library's own RCCL call on the compute stream.
    The device must already be bound (bind_device) — buffers created before that would sit on cuda:0 on every rank.
    force_comm: build the RCCL communicator and route opt.all_reduce() through it even when world == 1 (a one-rank
    all-reduce is the identity): the GPU tests use it to exercise the exact product path on a single-GPU box."""
    import torch.distributed as dist
    rank, local, world = dist_env()
    torch.cuda.set_device(local)
    if opt is not None and opt.device.type == "cuda" and opt.device.index not in (None, local):
        raise RuntimeError(f"rank {rank}: optimiser buffers are on {opt.device} but this process owns cuda:{local}; call "
                           "uia_hip.engine.bind_device(args) before building the model")
    if world > 1:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        uid = [ops.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        try:
            ops.comm_init(rank, world, uid[0])
        except Exception as e:                                 # surface WHICH rank failed: the others will be blocked in the init
            raise RuntimeError(f"rank {rank}/{world} (cuda:{local}): RCCL communicator init failed: {e}") from e
    elif force_comm and ops.comm_world_initialised() is False:
        ops.comm_init(0, 1, ops.comm_unique_id())
    if opt is not None:
        opt.world = world
        opt.collective = bool(force_comm)
        if world > 1:                                          # identical adapter weights on every rank
            dist.broadcast(opt.p, src=0)
    return rank, local, world


# ---- scalar agreements between ranks (control plane; torch.distributed, any backend).  Every rank must take the same branch
#      around a collective, otherwise the others block in it forever (a rank that `continue`s past opt.all_reduce()).
def all_ranks_agree(flag, device=None):
    """True iff `flag` is true on EVERY rank (single process: flag itself)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device if device is not None else _pg_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def sum_over_ranks(*values, device=None):
    """Element-wise sum of python scalars over the ranks (single process: the values)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tuple(float(v) for v in values)
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if device is not None else _pg_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(t.tolist())


def _pg_device():
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


class GatherFeaturesFn(torch.autograd.Function):
    """[B, E] local features → [world·B, E] in rank order (RCCL all-gather on the compute stream).  Every rank then evaluates the
    SAME global-batch loss, so the gradient of that loss with respect to this rank's rows is simply the local slice of the
    gathered gradient — no collective in the backward; the parameter gradients of the ranks add up in the usual all-reduce."""

    @staticmethod
    def forward(ctx, x, rank, world):
        x = x.contiguous()
        out = torch.empty(world * x.shape[0], x.shape[1], device=x.device, dtype=x.dtype)
        if world > 1 or (x.is_cuda and ops.comm_world_initialised() and ops.comm_world() == world):
            ops.allgather(x, out)          # also at world 1 once a communicator exists (init_data_parallel(force_comm=True)): the GPU tests
        else:                              # drive the very collective the multi-GPU job uses on a one-GPU box
            out.copy_(x)
        ctx.meta = (rank, x.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        rank, B = ctx.meta
        return g[rank * B:(rank + 1) * B].contiguous(), None, None


_SIDE_STREAM = {}


TEXT_STREAM_PRIORITY = int(os.environ.get("UIA_TEXT_PRIO", "0"))      # HIP stream priority of the text tower's stream (0 = default, -1 = high); experiment knob
SLICE_STREAM_PRIORITY = int(os.environ.get("UIA_SLICE_PRIO", "0"))    # ... of the image tower's second-slice stream(s)


def _side_stream(device):
    if device not in _SIDE_STREAM:
        _SIDE_STREAM[device] = torch.cuda.Stream(device=device, priority=TEXT_STREAM_PRIORITY)
    return _SIDE_STREAM[device]


_MB_STREAMS = {}


def _mb_streams(device, n):
    key = (device.index if device.index is not None else torch.cuda.current_device(), n)
    if key not in _MB_STREAMS:
        _MB_STREAMS[key] = [torch.cuda.Stream(device=device, priority=SLICE_STREAM_PRIORITY) for _ in range(n)]
    return _MB_STREAMS[key]


GRAD_RESID3 = True      # the residual gradient between a frozen tower's backward Functions as a three-byte tensor (functional.publish_grad3); A/B: bench.py --no-grad-resid3


def _hook_free(model):
    """No forward / backward hooks anywhere in the model: nothing but the towers' own Functions reads the tensors between them.  Looked at every step (a hook registered
    after the first step must switch the three-byte hand-off off): ~400 modules, three dict truth tests each."""
    for m in model.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return False
    return True


TEXT_SLICES = 1         # experiment knob: the (frozen, forward-only) text tower in this many equal slices on as many streams when the image tower is split
IMAGE_SLICES = 2        # experiment knob: > 2 cuts the images behind the first slice into IMAGE_SLICES - 1 equal slices on as many streams
IMAGE_SPLIT = 0.5       # fraction of a micro-batch's images that form the FIRST of two image-tower slices on two streams (0 = one slice); see contrastive_step.  A/B: bench.py --image-split 0


def _frozen_text(model):
    """Every trainable parameter lives in the image tower (conservative; looked at every step the caller passes inputs_ready — a tower unfrozen later must bring the wait back)."""
    return all(k.startswith("visual.") for k, p in model.named_parameters() if p.requires_grad)


def _wait_inputs(stream, cur, ready):
    """`stream` is about to read the caller's batch: behind the batch's own event when the caller handed one (a prefetching loader: the copy that filled the
    batch), else behind everything the caller's stream has enqueued."""
    if ready is not None:
        stream.wait_event(ready)
    else:
        stream.wait_stream(cur)


def contrastive_micro(model, criterion, images, ids, overlap_text=True, global_loss=False, streams=1, image_split=None, inputs_ready=False, ready=None, world=1,
                      loss_scale=1.0, features=None, opt=None):
    """Forward of both towers + InfoNCE + backward of ONE micro-batch: the gradients land in the trainable parameters' .grad (the flat optimiser's staging
    buffer).  Returns the (unscaled) loss, a device scalar; nothing is read on the host.  Call between begin_update()/end_update() — contrastive_step and
    ContrastiveLoop do.

    ready: an event after which `images` / `ids` are complete (DevicePrefetcher hands one per batch); with it, or with inputs_ready=True (the caller vouches
    that the batch was complete before the call), a FROZEN text tower's stream does not wait for the caller's stream — nothing it reads is written by the
    previous update — so, the host running ahead of the GPU, this micro-batch's text tower starts beside the previous backward's tail and the optimiser
    launches instead of behind them.
    features: optional (fi, ft) -> (fi, ft) applied before the loss (MetaCLIP's entry point normalises there, metaclip/finetune.py:92-97)."""
    mb = images.shape[0]
    if image_split is None:
        image_split = int(round(mb * IMAGE_SPLIT)) if mb >= 32 else 0
    sliced = streams > 1 and mb >= 2 * streams
    split = overlap_text and 0 < image_split < mb and not sliced
    ops.CHAINS = 3 if split else 1                               # forward AND backward of this micro-batch (ops.gemm's tail policy)
    cur = torch.cuda.current_stream()
    im, tk = images, ids
    ahead = (inputs_ready or ready is not None) and _frozen_text(model)
    if ready is not None:
        cur.wait_event(ready)
    if sliced:
        sts = _mb_streams(images.device, streams)
        bounds = [mb * s // streams for s in range(streams + 1)]
        fis, fts = [], []
        for s, st in enumerate(sts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                fis.append(model.encode_image(im[bounds[s]:bounds[s + 1]]))
                fts.append(model.encode_text(tk[bounds[s]:bounds[s + 1]]))
        for st, a, b in zip(sts, fis, fts):
            cur.wait_stream(st)
            a.record_stream(cur)
            b.record_stream(cur)
        fi, ft = torch.cat(fis, 0), torch.cat(fts, 0)
    elif split:
        # round 4: the image tower as TWO slices on two streams, the text tower whole on a third.  Every kernel of the tower is per row, per image or per
        # (image, head), so the slices are independent chains; a chain of dependent launches leaves CUs idle at every ragged last round and M tail (the
        # backward has no text tower beside it to fill them), two chains fill each other's.  One InfoNCE over all pairs; same loss and gradients (the weight
        # gradients meet through the same float atomics).  Halves measured best (41.43 -> 40.78 ms; 0.86 / 0.14 — a first slice of whole 256-tile rounds —
        # 41.09, thirds 41.16); slicing the text tower as well (streams = 3) is slower: its 768-tile launches are whole rounds already.
        side = _side_stream(images.device)
        if not ahead:
            side.wait_stream(cur)
        elif ready is not None:
            side.wait_event(ready)
        fts, tsides = None, []
        if TEXT_SLICES > 1:                                  # experiment knob: the text tower in equal slices on as many streams as well
            tb = [mb * j // TEXT_SLICES for j in range(TEXT_SLICES + 1)]
            tsides = [side] + _mb_streams(images.device, 8 + TEXT_SLICES - 1)[:TEXT_SLICES - 1]
            fts = []
            for j, st in enumerate(tsides):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    fts.append(model.encode_text(tk[tb[j]:tb[j + 1]]))
        else:
            with torch.cuda.stream(side):
                ft = model.encode_text(tk)
        cuts = [0, image_split] + ([image_split + (mb - image_split) * j // (IMAGE_SLICES - 1) for j in range(1, IMAGE_SLICES - 1)] if IMAGE_SLICES > 2 else []) + [mb]
        extra = _mb_streams(images.device, len(cuts) - 2)
        parts = [None] * (len(cuts) - 1)
        for j, st in enumerate(extra):                       # slices 1.. on their own streams, slice 0 on the caller's
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                parts[j + 1] = model.encode_image(im[cuts[j + 1]:cuts[j + 2]])
        parts[0] = model.encode_image(im[:cuts[1]])
        for j, st in enumerate(extra):
            cur.wait_stream(st)
            parts[j + 1].record_stream(cur)
        fi = torch.cat(parts, 0)
        if fts is not None:
            for st, f in zip(tsides, fts):
                cur.wait_stream(st)
                f.record_stream(cur)
            ft = torch.cat(fts, 0)
        else:
            cur.wait_stream(side)
            ft.record_stream(cur)
    elif overlap_text:
        side = _side_stream(images.device)
        if not ahead:
            side.wait_stream(cur)
        elif ready is not None:
            side.wait_event(ready)
        with torch.cuda.stream(side):
            ft = model.encode_text(tk)
        fi = model.encode_image(im)
        cur.wait_stream(side)
        ft.record_stream(cur)
    else:
        fi = model.encode_image(im)
        ft = model.encode_text(tk)
    if features is not None:
        fi, ft = features(fi, ft)
    if global_loss:
        rank, _, _ = dist_env()
        fi, ft = GatherFeaturesFn.apply(fi, rank, world), GatherFeaturesFn.apply(ft, rank, world)
    loss = criterion(fi, ft)
    if opt is not None:
        opt._staging_dirty = True                              # until accumulate() has folded (and zeroed) the staging buffer
    (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    if split:
        for st in _mb_streams(images.device, max(1, IMAGE_SLICES - 1)):
            cur.wait_stream(st)
    if sliced:
        for st in _mb_streams(images.device, streams):     # the adapters' weight gradients are side effects of the backward kernels (flat
            cur.wait_stream(st)                            # buffer, direct mode): autograd's own end-of-backward sync does not know them
    return loss.detach()


def begin_update(model, opt=None):
    """Per-update registries and the three-byte gradient hand-off (tokens are a contract of ONE update's forwards and backwards).  opt: the staging buffer (the
    .grad views) is zeroed when the previous micro-batch did not reach accumulate() — an exception mid-backward, a caller's own backward — so that stale finite
    gradients are never added to this update (accumulate() leaves the buffer zeroed and clears the flag)."""
    if opt is not None and getattr(opt, "_staging_dirty", False):
        opt.g.zero_()
    UF.clear_t_copies()
    UF.set_grad_resid3(GRAD_RESID3 and _hook_free(model))      # tokens are handed out only inside UF.linear_chain() scopes (the towers' own block loops)


def end_update():
    ops.CHAINS = 1        # also when a launch raised: the policy must not leak into the caller's next launches
    UF.set_grad_resid3(False)      # another loop in the process (segmentation heads on tapped blocks) starts without tokens


def contrastive_step(model, criterion, opt, images, ids, micro_batches=1, lr=None, overlap_text=True, global_loss=False, streams=1, image_split=None, inputs_ready=False,
                     lr_min=0.0, t_max=0):
    """One optimiser update: [encode -> InfoNCE -> backward -> guarded accumulate] x micro_batches (-> all-reduce) -> guarded clip + AdamW.  Returns the mean
    loss (device scalar).  NO host read anywhere: the reference's non-finite skip (finetune.py:281-285) is decided on the device (FlatAdapterOptimizer.accumulate /
    update).  This is the step bench.py times AND the step the fine-tune entry points run (ContrastiveLoop below calls the same two functions per loader batch).

    global_loss (opt-in, not the reference's semantics): the InfoNCE batch is the GLOBAL batch — features are all-gathered, every rank evaluates the same
    world·B x world·B loss, back-propagates the rows it owns, and the all-reduced parameter gradient is the gradient of that one loss (no 1/world scaling).
    Default: each rank's local loss, averaged (== the reference's gradient accumulation).

    overlap_text: the frozen text tower does not depend on the image tower, so it runs on a second HIP stream beside encode_image (same kernels, same
    results); the streams join before the loss.  With images of at least 32 per micro-batch the image tower itself runs as two half-batch slices on two
    streams (IMAGE_SPLIT): two chains of dependent launches fill each other's ragged last rounds and M tails.

    streams = S > 1: the batch is cut into S slices whose towers (forward AND backward: autograd runs a node on the stream of its forward) are enqueued on S
    HIP streams; the loss is still ONE InfoNCE over all B pairs."""
    begin_update(model, opt)
    total = None
    mb = images.shape[0] // micro_batches
    try:
        for i in range(micro_batches):
            loss = contrastive_micro(model, criterion, images[i * mb:(i + 1) * mb], ids[i * mb:(i + 1) * mb], overlap_text=overlap_text, global_loss=global_loss,
                                     streams=streams, image_split=image_split, inputs_ready=inputs_ready, world=opt.world, loss_scale=1.0 / micro_batches, opt=opt)
            opt.accumulate(loss)
            total = loss if total is None else total + loss
    finally:
        end_update()
    opt.update(lr=lr, lr_min=lr_min, t_max=t_max, grad_scale=dp_grad_scale(opt.world, global_loss))
    return total / micro_batches if micro_batches > 1 else total


class ContrastiveLoop:
    """The fine-tune entry points' training loop body (reference src/models/biomedclip/finetune.py:272-310) on the measured step: per LOADER batch one
    contrastive_micro + guarded accumulate, at every accumulation boundary (`(batch_idx + 1) % accumulation_steps == 0` or the loader's last batch, :297) one
    guarded update — no host read of the loss, so the host runs ahead of the GPU exactly as in bench.py.  The reference's decisions that depended on the
    loss value are taken on the device: a non-finite micro-batch contributes nothing, and when it is the boundary micro-batch the update check is skipped
    with it (the `continue` at :285), the schedule not advancing.  end_epoch() is the one host sync per epoch: it returns the figures the reference
    accumulated with loss.item() and the indices of the skipped batches for the reference's warning."""

    def __init__(self, model, criterion, opt, accumulation_steps=1, lr=1e-4, lr_min=0.0, total_updates=0, features=None, global_loss=False, discard_on_skip=False):
        self.model, self.criterion, self.opt, self.discard_on_skip = model, criterion, opt, discard_on_skip
        self.acc_steps, self.lr, self.lr_min, self.t_max = max(1, int(accumulation_steps)), lr, lr_min, int(total_updates)
        self.features, self.global_loss = features, global_loss
        self._open = False

    def begin_epoch(self, n_batches):
        self.n_batches = n_batches
        self.opt.start_log(n_batches)
        self._acc0 = getattr(self, "_acc_seen", 0)

    def micro(self, images, ids, batch_idx, ready=None, inputs_ready=False):
        if not self._open:
            begin_update(self.model, self.opt)
            self._open = True
        try:
            loss = contrastive_micro(self.model, self.criterion, images, ids, ready=ready, inputs_ready=inputs_ready, world=self.opt.world,
                                     loss_scale=1.0 / self.acc_steps, features=self.features, global_loss=self.global_loss, opt=self.opt)
            self.opt.accumulate(loss, log_index=batch_idx)
        except BaseException:
            end_update()
            self._open = False
            raise
        if ((batch_idx + 1) % self.acc_steps == 0) or (batch_idx + 1 == self.n_batches):
            end_update()
            self._open = False
            self.opt.update(lr=self.lr, lr_min=self.lr_min, t_max=self.t_max, grad_scale=dp_grad_scale(self.opt.world, self.global_loss), discard_on_skip=self.discard_on_skip)
        return loss

    def end_epoch(self):
        if self._open:
            end_update()
            self._open = False
        g = self.opt.read_guard()
        g["skipped_batches"] = [i for i, ok in enumerate(g["ok_log"] or []) if ok == 0]
        g["epoch_accumulated"] = g["accumulated"] - getattr(self, "_acc0", 0)      # finite micro-batches of THIS epoch (the counters on the device are cumulative)
        self._acc_seen = g["accumulated"]
        return g


def segmentation_step(model, criterion, opt, images, labels, input_ids=None, lr=None):
    """One iteration of the segmentation loops (reference src/models/clipseg/segmentation.py:138-148, biomedclip/segmentation.py:168-178): zero_grad -> forward ->
    DiceCE -> backward (-> all-reduce of the decoder / head gradients) -> AdamW with the caller's learning rate.  No gradient clipping, no finiteness check —
    the reference has neither here.  Nothing is read on the host.  The step `bench.py --config clipseg` times and the step the entry points run.  Returns
    (loss, logits), both on the device."""
    opt.zero_grad()
    preds = model(images, input_ids=input_ids) if input_ids is not None else model(images)
    loss = criterion(preds, labels)
    loss.backward()
    opt.all_reduce()
    opt.step(lr=lr)
    UF.clear_t_copies()
    return loss.detach(), preds.detach()


class DevicePrefetcher:
    """Double-buffered loader: a background thread pulls (images, texts) batches from a DataLoader, tokenises, stages them in a ring of pinned host buffers
    and copies them to a ring of device buffers on a COPY stream; the consumer gets (images, ids, event) — the batch is complete behind `event`, which is
    what lets contrastive_micro start a frozen text tower without waiting for the previous update (its `ready` argument).  A device slot is reused only behind
    an event the consumer's stream records when it asks for the next batch (every stream that read the slot has been joined into it by then)."""

    def __init__(self, loader, tokenizer, device, depth=2, second=None):
        """second: batch -> the tensor that travels beside the images (default: the token ids — batch[2] when the loader's workers tokenised, else
        tokenizer(batch[1]); the segmentation loaders pass the masks)."""
        import threading
        self.loader, self.tokenizer, self.device, self.depth, self.second = loader, tokenizer, torch.device(device), max(1, depth), second
        self._threading = threading
        self.wait_s = 0.0
        self._slots = None
        self._copy = torch.cuda.Stream(device=self.device)
        self._stop = threading.Event()
        self._live = None                                       # (producer thread, its queue) of the iteration in flight

    def __len__(self):
        return len(self.loader)

    def _make_slots(self, images, ids, staging=True):
        n = self.depth + 2
        self._slots = [{"h_im": None, "h_id": None,             # pinned staging buffers, made on first use (not needed when the loader's ring itself is pinned)
                        "d_im": torch.empty(images.shape, dtype=images.dtype, device=self.device), "d_id": torch.empty(ids.shape, dtype=ids.dtype, device=self.device),
                        "free": None, "copied": None} for _ in range(n)]

    def _producer(self, it, q):
        try:
            torch.cuda.set_device(self.device)
            ring = getattr(getattr(self.loader, "collate_fn", None), "release", None) and self.loader.collate_fn       # datasets.finetune.SharedBatchRing
            ring_dma = bool(ring is not None and ring.pin())    # host-to-device copies straight from the shared slot (registered as pinned memory)
            pending = []                                        # (event of the copy, ring slot): handed back to the workers once the copy has run
            k = 0
            for batch in it:
                slot = None
                if ring is not None and len(batch) == 3 and isinstance(batch[0], str) and batch[0] == ring.MARK:
                    slot = batch[1]
                    images, ids = ring.images[slot], ring.ids[slot]
                else:
                    images = batch[0]
                    if self.second is not None:
                        ids = self.second(batch)
                    else:
                        ids = batch[2] if len(batch) > 2 else self.tokenizer(list(batch[1]))      # a DataModule built with the tokenizer has tokenised in its workers
                n = images.shape[0]
                fits = (self._slots is not None and self._slots[0]["d_im"].shape[1:] == images.shape[1:] and self._slots[0]["d_id"].shape[1:] == ids.shape[1:]
                        and n <= self._slots[0]["d_im"].shape[0] and self._slots[0]["d_im"].dtype == images.dtype and self._slots[0]["d_id"].dtype == ids.dtype)
                if not fits:
                    self._make_slots(images, ids, staging=not (slot is not None and ring_dma))
                full = self._slots[k % len(self._slots)]
                k += 1
                sl = full if n == full["d_im"].shape[0] else self._view(full, n)      # a ragged last batch (validation / test splits) is a view of a full slot
                if slot is not None and ring_dma:
                    src_im, src_id = images, ids
                else:
                    if full.get("h_im") is None:
                        full["h_im"] = torch.empty(full["d_im"].shape, dtype=images.dtype, pin_memory=True)
                        full["h_id"] = torch.empty(full["d_id"].shape, dtype=ids.dtype, pin_memory=True)
                    if full["copied"] is not None:
                        full["copied"].synchronize()            # the pinned staging buffers are free once their copy has run
                    full["h_im"][:n].copy_(images)
                    full["h_id"][:n].copy_(ids)
                    src_im, src_id = full["h_im"][:n], full["h_id"][:n]
                    if slot is not None:
                        ring.release(slot)
                        slot = None
                with torch.cuda.stream(self._copy):
                    if full["free"] is not None:
                        self._copy.wait_event(full["free"])     # the consumer's last reader of this device slot
                    sl["d_im"].copy_(src_im, non_blocking=True)
                    sl["d_id"].copy_(src_id, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy)
                full["copied"] = ev
                if slot is not None:
                    pending.append((ev, slot))
                while pending and (pending[0][0].query() or len(pending) >= max(2, ring.slots - 2 * getattr(self.loader, "num_workers", 0) - 1)):
                    e, sidx = pending.pop(0)
                    e.synchronize()
                    ring.release(sidx)
                del batch, images, ids
                if not self._put(q, (sl, ev)):
                    break                                       # close(): the consumer is gone
            for e, sidx in pending:
                e.synchronize()
                ring.release(sidx)
            self._put(q, None)
        except BaseException as e:                              # surfaces in the consumer
            self._put(q, e)

    def _put(self, q, item):
        """q.put that gives up once close() was called (the consumer no longer takes anything)."""
        import queue
        while not self._stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    @staticmethod
    def _view(full, n):
        return {"d_im": full["d_im"][:n], "d_id": full["d_id"][:n], "_of": full}

    def close(self):
        """Stops the producer of the iteration in flight (an epoch abandoned by early stopping): no new batch is requested from the loader, ring slots
        already taken are handed back, the thread is joined.  Safe to call at any time, also twice."""
        live, self._live = self._live, None
        if live is None:
            return
        th, q = live
        self._stop.set()
        while th.is_alive():
            try:
                q.get(timeout=0.05)                             # frees the place a blocked put is waiting for
            except Exception:
                pass
            th.join(timeout=0.05)
        self._stop.clear()

    def __iter__(self):
        """Starts the producer NOW (not at the first next()): the first `depth` batches are loaded, staged and copied while the caller does something else."""
        import queue
        q = self._q = queue.Queue(maxsize=self.depth)
        first = self.loader.__dict__.pop("_uia_first_iter", None) if hasattr(self.loader, "__dict__") else None      # DataModule.start_workers() made it before the GPU was touched
        self.close()                                            # an abandoned earlier iteration
        th = self._threading.Thread(target=self._producer, args=(first if first is not None else iter(self.loader), q), daemon=True)
        th.start()
        self._live = (th, q)
        return self._consume(th, q)

    def _consume(self, th, q):
        prev = None
        while True:
            if prev is not None:                                # the consumer has enqueued everything that reads the previous slot; recorded BEFORE the queue
                ev = torch.cuda.Event()                         # frees a place: the producer may not reach this slot again until then
                ev.record(torch.cuda.current_stream(self.device))
                prev["free"] = ev
                prev = None
            t0 = time.perf_counter()
            item = q.get()
            self.wait_s += time.perf_counter() - t0              # time the consumer stood still for the loader (0 when the pipeline keeps up)
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            sl, ev = item
            prev = sl.get("_of", sl)
            yield sl["d_im"], sl["d_id"], ev
        th.join()
        if self._live is not None and self._live[0] is th:
            self._live = None
