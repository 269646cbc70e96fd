"""Autograd-level building blocks on top of uia_hip.ops (one process per GPU, current stream).

What lives here is orchestration only — which kernel runs on which buffer, what is stashed for the
backward — mirroring what PyTorch autograd does for the reference's modules:

  MonaFn        all four Mona variants                        (/root/reference/src/adapters/mona.py:96-487)
  VitBlockFn    pre-LN transformer block, dgrad-only backward  (src/third_party/openai_clip/model.py:177-202; timm Block)
                with optional LoRA on the attention projections (src/adapters/lora.py:54-199)
  post_ln_layer HF BERT layer, forward only (frozen text tower)
  patch_embed / head helpers

Residual stream, parameter gradients: fp32.  GEMM / attention operands: `T` = the compute dtype
(torch.bfloat16, or torch.float32 for the parity mode).
"""
import math
import os
import weakref

import time

import torch

from . import ops

_STATE = {"dtype": torch.bfloat16, "seed": 0x5EED, "calls": 0, "fwd_resid3": os.environ.get("UIA_FWD_RESID3", "1") != "0"}


def set_compute_dtype(dt):
    assert dt in (torch.bfloat16, torch.float32)
    _STATE["dtype"] = dt


def compute_dtype():
    return _STATE["dtype"]


def set_deferred_text_ln(flag):
    """Scheduling knob of the frozen text tower (results are bit-identical either way): True (default) keeps the fp32 residual of
    a post-LN sub-layer as the LayerNorm's INPUT + row statistics and lets the consuming GEMM epilogue normalise it; False makes
    every LayerNorm write its fp32 output as well."""
    _STATE["text_ln_deferred"] = bool(flag)


def set_block_resid3(flag):
    """bf16 mode with the LayerNorm fold (opt-in, default OFF: level with fp32 + T copy inside the step): inside a frozen pre-LN block the attention
    half's output x1 and its gradient dx1 are three-byte tensors (bf16 T copy + one low byte per element) between the GEMM epilogues and the
    LayerNorm backward."""
    _STATE["block_resid3"] = bool(flag)


def set_fwd_resid3(flag):
    """bf16 mode with the LayerNorm fold (default ON; A/B: bench.py --no-fwd-resid3): between a Mona adapter and the frozen block behind it the residual stream travels as a
    three-byte tensor in the FORWARD too — the bf16 T copy the block's QKV GEMM reads anyway plus one low byte per element — instead of fp32 + T copy: project2's epilogue
    writes 3 bytes per element instead of 6, the block's output projection reads 3 instead of 4 for its residual, and so does the LayerNorm backward that recomputes the
    block's first normalisation (the block saves 3 bytes per element for it instead of 4).  Only inside a tower's own block loop (linear_chain) and only when the next consumer
    is a plain frozen block: autograd carries a token (publish_fwd3), as for the three-byte gradients."""
    _STATE["fwd_resid3"] = bool(flag)


def set_text_resid3(flag):
    """bf16 mode with the LayerNorm fold (default on): the frozen post-LN text tower keeps its sub-layer sums as three-byte tensors — the bf16 T copy
    plus one low byte per element (uia_gemm_desc.resid_lo8 / out_lo8; 15 stored mantissa bits) — instead of fp32 + T copy.  False: fp32 sums."""
    _STATE["text_resid3"] = bool(flag)


def set_ln_fold(flag):
    """bf16 mode only (default on): the frozen pre-/post-LN blocks fold each LayerNorm into the GEMMs on either side of it — the producing
    epilogue writes the T copy of the raw rows and their (Σ, Σ²), the consuming GEMM runs on those raw rows with a weight pre-scaled by
    the LayerNorm weight and normalises its accumulators (uia_gemm_desc.rowsum_out / lnfold_*) — instead of a stand-alone LayerNorm pass
    over the rows.  fp32 mode always runs the LayerNorm kernels (exact parity path)."""
    _STATE["ln_fold"] = bool(flag)


def set_deterministic(flag):
    """True: nothing on the FORWARD path sums through float atomics — the split-K form of few-tile M tails (ops.TAIL_SPLIT_K, the ViT-L/14 + LoRA
    step's 128-row tails) is turned off, so features are bit-reproducible from run to run.  (Weight GRADIENTS still add up through float
    atomics in uia_wgrad and the Mona row kernels; their last bits vary either way.)"""
    _STATE["deterministic"] = bool(flag)
    ops.TAIL_SPLIT_K = not flag


def ln_fold_enabled(dt, rows=None):
    """rows: the fold is a large-batch optimisation and is only applied where its GEMMs run on the ring kernels (more than 2048 rows);
    below that nothing is gained.  (The row sums are 64-bit fixed-point integer atomics — exact and order-free, so the folded forward is
    bit-reproducible wherever no split-K tail runs — the headline step has none; the few-tile 128-row tails of ViT-L/14 (ops.TAIL_SPLIT_K)
    sum their K slices with float atomics: set_deterministic(True) removes them.  The first version of the fold used float atomics and was
    not reproducible at all: DESIGN.md §4.)"""
    return dt != torch.float32 and _STATE.get("ln_fold", True) and (rows is None or rows > _STATE.get("ln_fold_min_rows", 2048))


class LnFoldRangeError(FloatingPointError):
    """A GEMM that left row sums for a folded LayerNorm saw a non-finite or out-of-range partial sum: the LayerNorm statistics of that
    row — and everything downstream of it — are wrong."""


_FOLD_POLL = {}          # device index -> (pinned host word, event) of a read-back in flight
_FOLD_REPORTED = {}      # device index -> bits already reported


POLL_LAG = max(1, int(os.environ.get("UIA_POLL_LAG", "2")))     # steps between a guard word's copy to pinned memory and its examination on the host (poll_ln_flag, engine._poll_norm): with 1
                                                                # the host can never be more than ONE step ahead of the GPU, and any jitter of its enqueue time beyond the GPU's step time shows
                                                                # as a bubble (round 6: the CLIPSeg CLI, 8.3 ms steps enqueued in 6-7 ms beside a loader thread)


def reset_ln_flag(device=None):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    _FOLD_POLL.pop(key, None)
    _FOLD_REPORTED.pop(key, None)
    if key in ops._LN_FLAG:
        ops._LN_FLAG[key].zero_()


def poll_ln_flag(device=None, sync=False):
    """Reads the guard word of the folded LayerNorms (ops.ln_flag; include/uia_hip.h, uia_gemm_desc.ln_flag).  The fold's one assumption
    is roughly centred rows — bf16(x) rounds relative to |x|, so a row with |mean| >> std loses precision as a raw GEMM operand
    (DESIGN.md §4).  EVERY launch that writes or reads row sums checks the rows it touches on the device; this is the host side:
      bit 0 (a row with |mean| / std > ops.LN_FLAG_LIMIT): a warning, and the fold is switched off — the stand-alone LayerNorm kernels
            take over from the next forward;
      bit 1 (a non-finite or out-of-range partial sum): LnFoldRangeError — the step that produced it must not be trusted.
    sync=False (the training loops, once per step): no host synchronisation — the word is copied to pinned memory behind the step's
    kernels and the copy made POLL_LAG calls ago is examined, so a bad step is reported POLL_LAG (2) steps late; sync=True waits for this one."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    flag = ops._LN_FLAG.get(key)
    if flag is None:
        return 0
    seen = 0
    pending = _FOLD_POLL.setdefault(key, [])
    while pending and (sync or len(pending) >= POLL_LAG):       # the copy made POLL_LAG calls ago: the host may run that many steps ahead of the GPU before it waits here
        prev = pending.pop(0)
        t0 = time.perf_counter()
        prev[1].synchronize()
        _STATE["gpu_wait_s"] = _STATE.get("gpu_wait_s", 0.0) + time.perf_counter() - t0
        seen |= int(prev[0][0])
    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(flag, non_blocking=True)                         # the bits are sticky (no per-step fill launch): reset_ln_flag() clears them
    ev = torch.cuda.Event()
    ev.record()
    if sync:
        ev.synchronize()
        seen |= int(host[0])
    else:
        pending.append((host, ev))
    seen &= ~_FOLD_REPORTED.get(key, 0) | 2                       # a centring warning is given once; a range error every time it is seen
    _FOLD_REPORTED[key] = _FOLD_REPORTED.get(key, 0) | (seen & 1)
    if seen & 2:
        raise LnFoldRangeError("uia_hip: a LayerNorm row sum was non-finite or out of the fixed-point range (|Σ| or Σ² >= 5e8 in one column "
                               "block): the activations have diverged; the statistics of that row are not valid")
    if seen & 1 and _STATE.get("ln_fold", True):
        import warnings
        warnings.warn(f"uia_hip: LayerNorm fold switched off: a row with |mean| / std > {ops.LN_FLAG_LIMIT} would lose bf16 precision as a raw "
                      "GEMM operand; the LayerNorm kernels are used from the next forward on")
        set_ln_fold(False)
    return seen


def set_fp32_heads(flag):
    """bf16 mode (default on): the two feature heads — final LayerNorm of the CLS row + projection(s) — run on fp32 operands.  They are B x 768
    rows of work (nothing next to the towers), and they are the one place where a bf16 rounding lands on the features undamped: a feature is a
    768-term dot product whose terms largely cancel, so rounding its operands to 8 bits costs ≈ 2.8e-3 of the typical |feature| — about a
    quarter of the towers' whole error variance at B = 256 (tools/parity_at_bench_batch.py)."""
    _STATE["fp32_heads"] = bool(flag)


def head_dtype():
    return torch.float32 if _STATE.get("fp32_heads", True) else compute_dtype()


def set_unpad_text(flag):
    """Opt-in: the frozen text tower computes only the valid tokens of each caption (packed rows + per-caption attention) instead
    of all `context_length` positions.  Features are identical (padded positions never reach the pooled CLS row); what changes
    is the work executed, so the headline bench leaves it off."""
    _STATE["unpad_text"] = bool(flag)


def set_dropout_seed(seed):
    _STATE["seed"], _STATE["calls"] = int(seed), 0


def _next_seed():
    _STATE["calls"] += 1
    return (_STATE["seed"] * 0x9E3779B97F4A7C15 + _STATE["calls"] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


# ------------------------------------------------------------------------------------------------
# T copies of fp32 gradient tensors travel beside autograd: the kernel that produces a residual
# gradient also writes its T copy (operand of the consumer's first dgrad GEMM); the consumer looks
# it up by storage address instead of re-reading 4 bytes/element to cast.
_T_COPIES = {}


def publish_t_copy(g32, g_t):
    """The entry keeps a STRONG reference to the fp32 gradient: while it is registered its storage cannot be freed, so no
    later tensor can appear at the same address and pick up a stale copy (a registry keyed by address alone did exactly that
    when two backward passes ran without clear_t_copies() in between).  Entries are consumed by the first lookup; the
    training loops clear what was never looked up once per step, and the registry is bounded besides."""
    if g_t is not None and g_t.dtype != torch.float32:
        if len(_T_COPIES) >= 256:
            _T_COPIES.clear()
        _T_COPIES[g32.data_ptr()] = (g32, g32._version, g_t)


def t_copy_of(g32, dt, allow_kb=False):
    """Return a T copy of the fp32 gradient `g32` (the one its producer published, if it is still that tensor's content).
    allow_kb: the caller feeds the copy to a ring GEMM as its A operand and nothing else, so a K-blocked copy (3-D, ops.kb_empty) will do;
    every other caller gets a row-major [.., D] tensor (a K-blocked publication is ignored and the copy is cast afresh)."""
    if dt == torch.float32:
        return g32
    hit = _T_COPIES.pop(g32.data_ptr(), None)
    if (hit is not None and hit[0].shape == g32.shape and hit[0].stride() == g32.stride() and hit[1] == g32._version
            and hit[0].untyped_storage().data_ptr() == g32.untyped_storage().data_ptr() and hit[2].dtype == dt
            and (allow_kb or not ops.is_kb(hit[2]))):
        return hit[2]
    out = torch.empty(g32.shape, device=g32.device, dtype=dt)
    ops.cast(g32.contiguous(), out)
    return out


def clear_t_copies():
    _T_COPIES.clear()
    _ROWS.clear()
    _SUMS_ARENA.clear()
    _G3.clear()
    _F3.clear()


# ------------------------------------------------------------------------------------------------
# THREE-BYTE residual gradients between the backward Functions of a frozen tower (round 4).  The fp32 gradient of the residual stream is
# written by one row kernel (4 B + its 2-byte T copy) and read back by the next (4 B), three times per layer; as a three-byte tensor
# (ops / include/uia_hip.h: bf16 hi plane — the T copy itself — plus one low byte) the same hand-off is 3 B written and 3 B read.  autograd
# still wants a tensor of the gradient's shape: it gets a TOKEN — one NaN float expanded to that shape, no memory — and the planes travel
# in this registry under the token's address.  Only MonaFn and VitBlockFn produce and consume tokens, a producer hands one out only when its
# forward saw that the tensor it is the gradient of came straight (through views) from the other Function (so the token's only reader is
# that Function's backward), and anything else that reads a token reads NaN: a topology this does not cover fails loudly, not quietly.
# Opt-in per step (set_grad_resid3; engine.contrastive_step turns it on for hook-free towers).
_G3 = {}
_G3_POOL = {}
_G3_VIEWS = ("PermuteBackward0", "ViewBackward0", "UnsafeViewBackward0", "TransposeBackward0", "AliasBackward0", "ReshapeAliasBackward0")
_G3_PARTNERS = ("MonaFnBackward", "VitBlockFnBackward")


def set_grad_resid3(flag):
    _STATE["grad_resid3"] = bool(flag)


def grad_resid3_enabled():
    return bool(_STATE.get("grad_resid3", False))


# Forward twin of the gradient tokens below: a residual-stream VALUE handed from MonaFn to the next block's VitBlockFn as (hi plane, low bytes, row sums).
_F3 = {}
_F3_POOL = {}


def publish_fwd3(shape, device, hi, lo, sums):
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))
    pool = _F3_POOL.get(key)
    if pool is None:
        pool = _F3_POOL[key] = [torch.full((256,), float("nan"), device=device, dtype=torch.float32), 0]
    i = pool[1]
    pool[1] = (i + 1) % 256
    tok = pool[0][i:i + 1].view((1,) * len(shape)).expand(shape)
    if len(_F3) >= 256:
        _F3.clear()
    _F3[tok.data_ptr()] = (hi, lo, sums, tok.numel())
    return tok


def fwd3_of(x):
    """(hi, lo, sums) when x is a token of publish_fwd3 (consumed), else None.  Call BEFORE anything touches x's values (x.contiguous() would materialise NaNs)."""
    if not _F3 or x.dim() == 0 or any(st != 0 for st, n in zip(x.stride(), x.shape) if n > 1):
        return None
    hit = _F3.pop(x.data_ptr(), None)
    if hit is None or hit[3] != x.numel():
        return None
    return hit[0], hit[1], hit[2]


class linear_chain:
    """Scope of a tower's own block loop (forward_features / VisionTransformer.forward): inside it the residual stream is a plain chain block -> adapter -> block,
    every block output has exactly ONE consumer, so a three-byte gradient token can stand in for its gradient.  Code that walks the blocks itself and taps
    intermediate outputs (FPN heads, CLIPSeg's extract layers) never enters the scope and gets fp32 residual gradients (a token with a second
    consumer would be ADDED to a real gradient by autograd)."""

    def __enter__(self):
        _STATE["chain_depth"] = _STATE.get("chain_depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _STATE["chain_depth"] -= 1
        _STATE["fwd3_next_plain"] = False
        return False

    @staticmethod
    def next_is_plain_block(flag):
        """The tower's loop says, before it runs block i, whether block i + 1 exists and is a plain frozen block (VitBlockFn): only then may block i's adapter hand its
        output over as a three-byte forward token (set_fwd_resid3)."""
        _STATE["fwd3_next_plain"] = bool(flag)


def hook_free(*modules):
    """No forward / forward-pre / backward hook on any of the modules or their submodules, and no global module hook: nothing but the towers' own Functions
    reads the tensors handed between them.  A three-byte forward token is a stride-0 NaN placeholder to everyone but its consumer (an activation
    tap or Grad-CAM hook on a block would read NaNs; a hook that returns a new tensor would feed them into the next block)."""
    import torch.nn.modules.module as _mm
    if _mm._global_forward_hooks or _mm._global_forward_pre_hooks or _mm._global_backward_hooks or getattr(_mm, "_global_backward_pre_hooks", None) \
            or getattr(_mm, "_global_forward_hooks_always_called", None):
        return False
    for mod in modules:
        for m in mod.modules():
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
                return False
    return True


def _g3_partner_feeds(x):
    """True when x is the output of a MonaFn / VitBlockFn seen through view nodes only: the gradient this Function returns for x goes to that Function's backward."""
    if not grad_resid3_enabled() or _STATE.get("chain_depth", 0) <= 0:
        return False
    fn = x.grad_fn
    for _ in range(8):
        if fn is None:
            return False
        name = type(fn).__name__
        if name in _G3_PARTNERS:
            return True
        if name not in _G3_VIEWS or len(fn.next_functions) != 1:
            return False
        fn = fn.next_functions[0][0]
    return False


def publish_grad3(shape, device, hi, lo):
    """Register the planes of a three-byte gradient and return the token autograd carries in its place."""
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))
    pool = _G3_POOL.get(key)
    if pool is None:
        pool = _G3_POOL[key] = [torch.full((64,), float("nan"), device=device, dtype=torch.float32), 0]
    i = pool[1]
    pool[1] = (i + 1) % 64
    tok = pool[0][i:i + 1].view((1,) * len(shape)).expand(shape)
    if len(_G3) >= 64:
        _G3.clear()
    _G3[tok.data_ptr()] = (hi, lo, tok.numel())
    return tok


def grad3_of(g):
    """(hi, lo) when g is a token of publish_grad3 (consumed), else None.  Call BEFORE anything touches g's values."""
    if not _G3 or g.dim() == 0 or any(st != 0 for st, n in zip(g.stride(), g.shape) if n > 1):      # (a dimension of size 1 keeps whatever stride it had: a one-image slice)
        return None
    hit = _G3.pop(g.data_ptr(), None)
    if hit is None or hit[2] != g.numel():
        return None
    return hit[0], hit[1]


def grad3_decode(hi, lo):
    """fp32 values of a three-byte gradient (slow path of a consumer that cannot take the planes: torch ops)."""
    return ops.three_byte_to_float(hi, lo).contiguous()


# Zeroed [M, 2] row-sum buffers for the producers of folded LayerNorms: slices of one arena that a single fill zeroes (a block needs two or
# three of them per forward; 36 fills of 400 KB per step otherwise).  A slice is handed out once; the arena is dropped with the other
# per-step registries (clear_t_copies) or when it runs out, and a new one is zeroed on the next request.  One arena per STREAM: the fill
# that zeroes it and the atomics that land in its slices are ordered by that stream only (contrastive_step(overlap_text=True) runs the
# text tower on a second stream; a shared arena would hand it slices with no event between the fill and their first use).
_SUMS_ARENA = {}


def _stream_key():
    return ops.raw_stream()


def zero_sums(M, device):
    key = _stream_key()
    a = _SUMS_ARENA.get(key)
    if a is None or a[0].shape[1] != M or a[0].device != device or a[1] >= a[0].shape[0]:
        a = [torch.zeros(48, M, 2, device=device, dtype=torch.int64), 0]
        _SUMS_ARENA[key] = a
    a[1] += 1
    return a[0][a[1] - 1]


# The forward twin of the registry above, one slot deep: the GEMM that produces a residual-stream tensor (Mona project2, a block's fc2)
# can leave the T copy of its rows and their (Σ, Σ²) for the LayerNorm folded into the next block's first GEMM.
_ROWS = {}          # one slot per stream (the producer and the consumer of a hand-off run on the same stream)


def publish_rows(x32, x_t, sums):
    _ROWS[_stream_key()] = (x32, x32._version, x_t, sums)


def take_rows(x32, dt):
    hit = _ROWS.pop(_stream_key(), None)
    if (hit is not None and hit[0].data_ptr() == x32.data_ptr() and hit[0].numel() == x32.numel() and hit[1] == x32._version
            and hit[0].untyped_storage().data_ptr() == x32.untyped_storage().data_ptr() and x32.is_contiguous() and hit[2].dtype == dt):
        return hit[2], hit[3]
    return None


# ------------------------------------------------------------------------------------------------
class _PackedParam:
    """All GEMM-operand forms of one small trainable matrix, refreshed in place by uia_pack_weights.  With `pads` = (rows_pad, cols_pad) the
    forms hold the matrix zero-padded to that extent (a LoRA factor's rank 16 -> 64): the buffers are zeroed once and every refresh writes the
    parameter's own elements only."""
    __slots__ = ("ref", "version", "epoch", "dt", "row", "row_kb", "tr", "tr_kb", "fwd", "bwd", "src_ptr", "pads", "extra")

    def __init__(self, p, dt, pads=None):
        R, Cc = p.shape
        RP, CP = pads if pads is not None else (R, Cc)
        g = 64 // torch.empty(0, dtype=dt).element_size()
        mk = (lambda *shape: torch.zeros(shape, device=p.device, dtype=dt)) if pads is not None else (lambda *shape: torch.empty(shape, device=p.device, dtype=dt))
        self.ref, self.version, self.epoch, self.dt, self.src_ptr, self.pads = weakref.ref(p), -1, -1, dt, 0, pads
        self.extra = []                    # further destinations refreshed with the parameter: (K-blocked view inside another buffer, rows_pad, cols_pad, scale)
        self.row, self.tr = mk(RP, CP), mk(CP, RP)
        self.row_kb = mk(CP // g, RP, g) if CP % g == 0 else None
        self.tr_kb = mk(RP // g, CP, g) if RP % g == 0 else None
        self.fwd, self.bwd = ops.PackedW(self.row, self.row_kb), ops.PackedW(self.tr, self.tr_kb)

    def entries(self):
        src = self.ref().detach()
        ent = (src, self.row, self.row_kb, self.tr, self.tr_kb)
        out = [ent if self.pads is None else ent + ((self.pads[0], self.pads[1], 1.0),)]
        for view, rp, cp, scale in self.extra:
            out.append((src, None, ops.RawDest(view), None, None, (rp, cp, scale)))
        return out


class WeightCache:
    """T copies (and transposes, for dgrad-as-TN) of fp32 parameters, refreshed when the parameter's
    version counter changes (frozen weights are converted exactly once).

    Small trainable matrices (the adapters' projections) take a batched path: their row, transposed and K-blocked forms live in
    persistent buffers that ONE uia_pack_weights launch rewrites after every optimiser step (bump()); before that every Mona
    layer paid two casts, two transposes and two layout copies per step, 73 launches of ~5 us for 12 layers."""
    PACK_MAX_ELEMS = 1 << 20

    def __init__(self):
        self._c = {}
        self.epoch = 0
        self._packed = {}                  # id(p) -> _PackedParam
        self._ext_of = {}                  # storage pointer of an ExtW's K-blocked buffer -> weakref(ExtW): the pack launch rewrites its factor columns
        self._table = None                 # (device table, n, max_elems, dtype, [entries kept alive])

    def _refreshed(self, items):
        """one event for everything the pack launch just wrote: the operand forms of `items` and the factor columns inside the ExtW buffers"""
        ready = ops.Ready()
        for it in items:
            it.fwd.refreshed(ready)
            it.bwd.refreshed(ready)
            for view, _, _, _ in it.extra:
                ext = self._ext_of.get(view.untyped_storage().data_ptr())
                ext = ext() if ext is not None else None
                if ext is not None:
                    ext.ready = ready

    def _repack(self, items):
        dt, dev = items[0].dt, items[0].row.device
        entries = [e for it in items for e in it.entries()]
        table, n, mx = ops.pack_table(entries, dev)
        ops.pack_weights(table, n, mx, dt)
        for it in items:
            it.version, it.epoch, it.src_ptr = it.ref()._version, self.epoch, it.ref().data_ptr()
        self._refreshed(items)
        return table, n, mx, dt, entries

    def bump(self):
        """Trainable parameters were updated behind autograd's back (fused optimiser kernel)."""
        self.epoch += 1
        if len(self._c) > 8192:
            alive = lambda r: all(x() is not None for x in r) if isinstance(r, tuple) else r() is not None
            self._c = {k: v for k, v in self._c.items() if alive(v[1])}
        live = [it for it in self._packed.values() if it.ref() is not None]
        if len(live) != len(self._packed):
            self._packed = {id(it.ref()): it for it in live}
            self._table = None
        if not live:
            return
        groups = {}
        for it in live:
            groups.setdefault((it.dt, it.row.device), []).append(it)
        # One resident descriptor table per (dtype, device) group: while a group holds the same parameters at the same addresses as last step its table is still
        # right, and the step pays one pack launch per group — not a host-built table and a pageable host-to-device copy, which WAITS for the stream (seven of them
        # per CLIPSeg step, fp32 heads beside the bf16 decoder: 1.6 ms of a host-bound 10 ms step).
        tables = self._table if isinstance(self._table, dict) else {}
        fresh = {}
        for key, items in groups.items():
            sig = (sum(1 + len(it.extra) for it in items), tuple(id(it) for it in items))
            hit = tables.get(key)
            if hit is not None and hit[5] == sig and all(it.src_ptr == it.ref().data_ptr() for it in items):
                table, n, mx, dt, _, _ = hit                  # same set as last step: the resident table is still right
                ops.pack_weights(table, n, mx, dt)
                for it in items:
                    it.version, it.epoch = it.ref()._version, self.epoch
                self._refreshed(items)
                fresh[key] = hit
            else:
                fresh[key] = self._repack(items) + (sig,)
        self._table = fresh

    def _packed_get(self, p, dt, transpose, pads=None):
        it = self._packed.get(id(p))
        want = tuple(pads) if pads is not None else tuple(p.shape)
        if it is None or it.ref() is not p or it.dt != dt or it.row.device != p.device or tuple(it.row.shape) != want or it.pads != pads:
            old = it
            it = _PackedParam(p, dt, pads)
            if old is not None and old.ref() is p and old.dt == dt and old.row.device == p.device:
                # the same parameter asked for in another padded form: the factor columns it feeds inside cached ExtW buffers stay its
                # destinations (dropping them left a stale s·B in the K-extension forward), minus those whose ExtW is gone
                it.extra = [e for e in old.extra if (self._ext_of.get(e[0].untyped_storage().data_ptr()) or (lambda: None))() is not None]
            self._packed[id(p)] = it
            self._table = None
        if it.version != p._version or it.epoch != self.epoch:
            self._repack([it])                                # first use, or the parameter changed outside the optimiser step
        return it.bwd if transpose else it.fwd

    def get_lnfold(self, w, b, ln_w, ln_b, dt):
        """Operands of a Linear with the LayerNorm in front of it folded in (frozen weights; set_ln_fold): (W' = w·ln_w[None, :] in `dt` as a
        PackedW, colsum[n] = Σ_k W'[n][k] of the ROUNDED operand in fp32, bias' = b + w @ ln_b in fp32).  One-time weight preparation with
        torch ops, cached per (weight, LayerNorm weight) and refreshed when any of the four tensors changes."""
        key = (id(w), id(ln_w), dt, "lnfold")
        vers = (w._version, ln_w._version, ln_b._version, -1 if b is None else b._version, w.device)
        hit = self._c.get(key)
        if hit is not None and hit[0] == vers and hit[1]() is w and hit[3]() is ln_w:
            return hit[2]
        with torch.no_grad():
            w32 = w.detach().float()
            wf = (w32 * ln_w.detach().float()[None, :]).to(dt).contiguous()
            colsum = wf.float().sum(1).contiguous()
            bias = (w32 @ ln_b.detach().float())
            if b is not None:
                bias = bias + b.detach().float()
            bias = bias.contiguous()
        out = (ops.PackedW(wf), colsum, bias)       # constructed last: its event covers colsum and bias too
        self._c[key] = (vers, weakref.ref(w), out, weakref.ref(ln_w))
        return out

    def get_cat(self, ws, dt, transpose=False):
        """The frozen weights `ws` ([N_i, K] each) stacked along N as ONE GEMM operand ([ΣN_i, K]; transpose: [K, ΣN_i] for the data gradient):
        the q, k, v projections of a block run as one launch.  Converted once (frozen weights), refreshed if any of them changes."""
        key = (tuple(id(w) for w in ws), dt, transpose, "cat")
        vers = tuple(w._version for w in ws)
        hit = self._c.get(key)
        if hit is not None and hit[0] == vers and all(r() is w for r, w in zip(hit[1], ws)) and hit[2].device == ws[0].device:
            return hit[2]
        with torch.no_grad():
            cat = torch.cat([w.detach().float() for w in ws], 0)
            src = cat.t().contiguous() if transpose else cat
            out = ops.PackedW(src.to(dt).contiguous())
        self._c[key] = (vers, tuple(weakref.ref(w) for w in ws), out)
        return out

    def get_lora_ext(self, ws, Bs, scaling, dt, rp=64):
        """[W | s·B] over K + rp columns, K-blocked, for the K-extension form of the LoRA rank update (uia_gemm_desc.A2 / K2): the frozen weights `ws`
        stacked along N ([ΣN_i, K], converted once), behind them the factors `Bs` ([N_i, r] each, rank zero-padded to rp, scaled by s) in the last rp / g column
        blocks.  The factor part is a destination of the batched uia_pack_weights launch (refreshed with the parameters, like their other forms)."""
        key = (tuple(id(w) for w in ws), tuple(id(b) for b in Bs), dt, float(scaling), "ext")
        vers = tuple(w._version for w in ws)
        hit = self._c.get(key)
        pads = [(b.shape[0], max(rp, b.shape[1])) for b in Bs]
        if hit is not None and hit[0] == vers and all(r() is t for r, t in zip(hit[1], tuple(ws) + tuple(Bs))) and hit[2].device == ws[0].device:
            for b, pd in zip(Bs, pads):
                self._packed_get(b, dt, False, pd)                 # refreshes every destination of a factor that changed outside the optimiser step
            return hit[2]
        g = 64 // torch.empty(0, dtype=dt).element_size()
        Ntot, K = sum(w.shape[0] for w in ws), ws[0].shape[1]
        assert K % g == 0 and rp % g == 0 and all(w.shape[1] == K for w in ws) and all(b.shape[0] == w.shape[0] for b, w in zip(Bs, ws))
        with torch.no_grad():
            kb = torch.zeros((K + rp) // g, Ntot, g, device=ws[0].device, dtype=dt)
            kb[:K // g] = torch.cat([w.detach().float() for w in ws], 0).to(dt).view(Ntot, K // g, g).permute(1, 0, 2)
        off = 0
        for b, pd in zip(Bs, pads):
            self._packed_get(b, dt, False, pd)
            it = self._packed[id(b)]
            it.extra = [e for e in it.extra if e[0].untyped_storage().data_ptr() != kb.untyped_storage().data_ptr()]
            it.extra.append((kb[K // g:, off:off + b.shape[0], :], Ntot, rp, float(scaling)))
            self._table = None
            self._repack([it])
            off += b.shape[0]
        out = ops.ExtW(kb, Ntot, K + rp, rp)
        self._ext_of = {k: r for k, r in self._ext_of.items() if r() is not None}
        self._ext_of[kb.untyped_storage().data_ptr()] = weakref.ref(out)
        self._c[key] = (vers, tuple(weakref.ref(t) for t in tuple(ws) + tuple(Bs)), out)
        return out

    def get(self, p, dt, transpose=False, pad_rows_to=None, pad_cols_to=None):
        padded = pad_rows_to is not None or pad_cols_to is not None
        if (p.requires_grad and (dt != torch.float32 or padded) and p.dim() == 2 and p.is_cuda
                and p.dtype == torch.float32 and p.is_contiguous() and 0 < p.numel() <= self.PACK_MAX_ELEMS):
            pads = (max(p.shape[0], pad_rows_to or 0), max(p.shape[1], pad_cols_to or 0)) if padded else None
            return self._packed_get(p, dt, transpose, pads)
        key = (id(p), dt, transpose, pad_rows_to, pad_cols_to)
        hit = self._c.get(key)
        if hit is not None and hit[0] == p._version and hit[1]() is p and hit[2].device == p.device and (not p.requires_grad or hit[3] == self.epoch):
            return hit[2]
        src = p.detach()
        if src.dim() != 2:
            src = src.reshape(src.shape[0], -1)
        src = src.contiguous().float()
        if pad_rows_to is not None and src.shape[0] < pad_rows_to:      # LoRA rank → 64 (zero rows change nothing)
            src = torch.cat([src, src.new_zeros(pad_rows_to - src.shape[0], src.shape[1])], 0)
        if pad_cols_to is not None and src.shape[1] < pad_cols_to:      # contraction dim → the GEMM's K granule (zero columns change nothing)
            src = torch.cat([src, src.new_zeros(src.shape[0], pad_cols_to - src.shape[1])], 1)
        if transpose:
            out = torch.empty(src.shape[1], src.shape[0], device=src.device, dtype=dt)
            ops.transpose_cast(src, out)
        elif dt == torch.float32:
            out = src
        else:
            out = torch.empty(src.shape, device=src.device, dtype=dt)
            ops.cast(src, out)
        out = ops.PackedW(out)                                 # the ring GEMMs take the K-blocked twin, built on first use
        self._c[key] = (p._version, weakref.ref(p), out, self.epoch)
        return out


WEIGHTS = WeightCache()


def _empty(shape, dt, like):
    return torch.empty(shape, device=like.device, dtype=dt)


def _act(M, K, dt, like, n_consumer):
    """Storage of an [M, K] activation that one GEMM epilogue writes and one GEMM with n_consumer columns reads: K-blocked
    ([K/g, M, g], ops.kb_empty) when both launches run on the ring kernels, plain row-major otherwise."""
    if ops.kb_ok(M, n_consumer, K, dt):
        return ops.kb_empty(M, K, dt, like.device)
    return torch.empty((M, K), device=like.device, dtype=dt)


def _attn_act(M, K, dt, like, n_consumer):
    """An activation the attention kernels write and a GEMM reads (forward output, fused dq/dk/dv): their K-blocked stores exist on the
    bf16 path only."""
    return _act(M, K, dt, like, n_consumer) if dt == torch.bfloat16 else torch.empty((M, K), device=like.device, dtype=dt)


def _as_act(buf, M, K, dt, n_consumer):
    """The same, re-using the storage of a dead [M, K] buffer (either layout) of the same dtype."""
    if ops.kb_ok(M, n_consumer, K, dt):
        return buf if ops.is_kb(buf) else ops.KBlocked.over(buf.view(M, K))
    return buf.as_rows() if ops.is_kb(buf) else buf


# ================================================================================================ Mona
MONA_PARAM_ORDER = ("gamma", "gammax", "project1.weight", "project1.bias", "project2.weight", "project2.bias", "norm.weight", "norm.bias",
                    "adapter_conv.conv1.weight", "adapter_conv.conv1.bias", "adapter_conv.conv2.weight", "adapter_conv.conv2.bias",
                    "adapter_conv.conv3.weight", "adapter_conv.conv3.bias", "adapter_conv.projector.weight", "adapter_conv.projector.bias",
                    "adapter_conv.freq_filter", "adapter_conv.noise_estimator.1.weight", "adapter_conv.noise_estimator.1.bias",
                    "adapter_conv.noise_estimator.3.weight", "adapter_conv.noise_estimator.3.bias")
_SPATIAL_MAP = {"adapter_conv.conv1.weight": "conv1_w", "adapter_conv.conv1.bias": "conv1_b", "adapter_conv.conv2.weight": "conv2_w",
                "adapter_conv.conv2.bias": "conv2_b", "adapter_conv.conv3.weight": "conv3_w", "adapter_conv.conv3.bias": "conv3_b",
                "adapter_conv.projector.weight": "proj_w", "adapter_conv.projector.bias": "proj_b", "adapter_conv.freq_filter": "freq",
                "adapter_conv.noise_estimator.1.weight": "ne1_w", "adapter_conv.noise_estimator.1.bias": "ne1_b",
                "adapter_conv.noise_estimator.3.weight": "ne3_w", "adapter_conv.noise_estimator.3.bias": "ne3_b"}


class MonaFn(torch.autograd.Function):
    """y = x + project2(drop(gelu(spatial(project1(LN(x)·γ + x·γx)))))  on batch-first x [B, N, D] fp32."""

    @staticmethod
    def forward(ctx, x, variant, hw, p_drop, keep_mask, names, direct, *params):
        P = dict(zip(names, params))
        B, N, D = x.shape
        h, w = hw
        assert N == 1 + h * w, f"Mona expects 1+h*w tokens, got N={N}, hw={hw}"
        dt = compute_dtype()
        x = x.contiguous()
        M = B * N
        bott = P["project1.weight"].shape[0]
        if ops.mona_fused_ok(dt, D, h, w, bott):
            # the whole adapter in one launch (csrc/mona_fused.hip): u, t and d never travel through HBM between the stages
            sp = {_SPATIAL_MAP[k]: v.detach().contiguous() for k, v in P.items() if k in _SPATIAL_MAP}
            seed = _next_seed() if (p_drop > 0 and keep_mask is None) else 0
            train = any(ctx.needs_input_grad)
            u = _empty((M, D), dt, x) if train else None
            t = _empty((M, bott), dt, x) if train else None
            d = _empty((M, bott), dt, x) if train else None
            w1, w2 = WEIGHTS.get(P["project1.weight"], dt), WEIGHTS.get(P["project2.weight"], dt)
            y = torch.empty_like(x)
            fold = ln_fold_enabled(dt, M)
            y_t = _act(M, D, dt, x, 3 * D) if fold else None
            sums = zero_sums(M, x.device) if fold else None
            ops.mona_fused_fwd(variant, B, h, w, x, P["norm.weight"], P["norm.bias"], P["gamma"], P["gammax"], w1.row, P["project1.bias"], w2.row,
                               P["project2.bias"], sp, y, y_t=y_t, rowsum=sums, u_out=u, t_out=t, d_out=d, p_drop=p_drop, seed=seed, keep_mask=keep_mask)
            if fold:
                publish_rows(y, y_t, sums)
            if train:
                ctx.save_for_backward(x, u, t, d, keep_mask if keep_mask is not None else x.new_empty(0), *params)
            ctx.meta = (variant, hw, p_drop, seed, names, keep_mask is not None)
            ctx.direct_params = tuple(params) if direct else None
            ctx.g3_out = dt == torch.bfloat16 and _g3_partner_feeds(x)
            return y
        u = _empty((M, D), dt, x)
        w1 = WEIGHTS.get(P["project1.weight"], dt)
        t = _empty((M, bott), dt, x)
        if ops.MONA_PRE_FWD_T and x.is_cuda and dt == torch.bfloat16 and D == 768 and bott == 64:
            # project1 inside the row kernel: t from the u tile in LDS, u itself still written for the backward's weight gradient
            ops.mona_pre_fwd(x, P["norm.weight"], P["norm.bias"], P["gamma"], P["gammax"], u, proj1=(w1.row if isinstance(w1, ops.PackedW) else w1, P["project1.bias"], t))
        else:
            ops.mona_pre_fwd(x, P["norm.weight"], P["norm.bias"], P["gamma"], P["gammax"], u)
            ops.gemm(u, w1, bias=P["project1.bias"], out_t=t)
        sp = {_SPATIAL_MAP[k]: v.detach().contiguous() for k, v in P.items() if k in _SPATIAL_MAP}
        d = _empty((M, bott), dt, x)
        seed = _next_seed() if (p_drop > 0 and keep_mask is None) else 0
        ops.mona_spatial_fwd(variant, B, h, w, t, sp, d, p_drop=p_drop, seed=seed, keep_mask=keep_mask)
        w2 = WEIGHTS.get(P["project2.weight"], dt)
        y = torch.empty_like(x)
        if ln_fold_enabled(dt, M):             # the next block's first LayerNorm is folded into its QKV GEMM: leave it the T rows and their sums
            y_t, sums = _act(M, D, dt, x, 3 * D), zero_sums(M, x.device)          # read by the next block's QKV GEMM only
            if (dt == torch.bfloat16 and _STATE.get("fwd_resid3", True) and _STATE.get("fwd3_next_plain", False) and _STATE.get("chain_depth", 0) > 0 and M > 2048
                    and x.is_cuda):
                # the next block takes the sum as a THREE-BYTE tensor: its hi plane is the T copy, one low byte per element beside it; no fp32 rows are written
                y_lo = torch.empty(M, D, device=x.device, dtype=torch.int8)
                ops.gemm(d, w2, bias=P["project2.bias"], resid=x.view(M, D), out_t=y_t, out_lo=y_lo, rowsum=sums)
                y = publish_fwd3(x.shape, x.device, y_t, y_lo, sums)
            else:
                ops.gemm(d, w2, bias=P["project2.bias"], resid=x.view(M, D), out32=y.view(M, D), out_t=y_t, rowsum=sums)
                publish_rows(y, y_t, sums)
        else:
            ops.gemm(d, w2, bias=P["project2.bias"], resid=x.view(M, D), out32=y.view(M, D))
        ctx.save_for_backward(x, u, t, d, keep_mask if keep_mask is not None else x.new_empty(0), *params)
        ctx.meta = (variant, hw, p_drop, seed, names, keep_mask is not None)
        ctx.direct_params = tuple(params) if direct else None      # the Parameter objects themselves: .grad is looked up at BACKWARD time
        ctx.g3_out = dt == torch.bfloat16 and _g3_partner_feeds(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        variant, hw, p_drop, seed, names, has_mask = ctx.meta
        x, u, t, d, keep_mask, *params = ctx.saved_tensors
        keep_mask = keep_mask if has_mask else None
        P = dict(zip(names, params))
        B, N, D = x.shape
        h, w = hw
        M, dt = B * N, u.dtype
        bott = t.shape[1]
        g3 = grad3_of(dy)                                    # a three-byte gradient from the block behind this adapter: (T copy, low bytes)
        fuse_du = x.is_cuda and ops.mona_pre_bwd_du_ok(M, D, bott, dt)
        if g3 is not None and ops.is_kb(g3[0]):
            dy, g3 = grad3_decode(*g3).view(B, N, D), None   # a K-blocked hi plane is no operand of this backward's launches: back to fp32 (torch ops; not a path the towers take)
        if g3 is not None:
            dy_t = g3[0].view(M, D)
        else:
            dy = dy.contiguous()
            dy_t = t_copy_of(dy, dt).view(M, D)
        # Direct mode: accumulate straight into the flat-buffer .grad views (no fills, no adds, nothing returned to autograd).
        # It is an explicit opt-in of FlatAdapterOptimizer (engine.py marks its parameters), and the views are re-read HERE,
        # at backward time: if the caller dropped or replaced them in between (optimizer.zero_grad(set_to_none=True),
        # torch.autograd.grad, a GradScaler-style consumer) the gradients are returned to autograd like any other Function.
        direct = ctx.direct_params is not None and all(_is_flat_grad(p) for p in ctx.direct_params)
        G = ({k: p.grad for k, p in zip(names, ctx.direct_params)} if direct
             else {k: torch.zeros_like(v, dtype=torch.float32) for k, v in P.items()})
        # project2: dd = dy·W2 ; dW2 = dyᵀ·d ; db2 = Σ dy
        w2t = WEIGHTS.get(P["project2.weight"], dt, transpose=True)          # [bott, D]
        dd = _empty((M, bott), dt, x)
        side = _wgrad_side_stream(x.device) if _STATE.get("wgrad_side_stream", False) else None
        if side is not None:
            # the two weight gradients of the adapter depend on nothing the data-gradient chain produces later: on a second stream they
            # run beside its launches (whose lockstep phases and one-round kernels leave CUs and HBM idle)
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.wgrad(dy_t, d, G["project2.weight"], G["project2.bias"])
            dy_t.record_stream(side)
            d.record_stream(side)
        ops.gemm(dy_t, w2t, out_t=dd)
        if side is None:
            ops.wgrad(dy_t, d, G["project2.weight"], G["project2.bias"])
        # spatial
        sp = {_SPATIAL_MAP[k]: v.detach().contiguous() for k, v in P.items() if k in _SPATIAL_MAP}
        sg = {_SPATIAL_MAP[k]: G[k] for k in P if k in _SPATIAL_MAP}
        dtt = _empty((M, bott), dt, x)
        ops.mona_spatial_bwd(variant, B, h, w, t, sp, dd, dtt, sg, p_drop=p_drop, seed=seed, keep_mask=keep_mask)
        # project1: du = dt·W1 ; dW1 = dtᵀ·u ; db1 = Σ dt
        w1t = WEIGHTS.get(P["project1.weight"], dt, transpose=True)          # [D, bott]
        du = None if fuse_du else _empty((M, D), dt, x)                      # fuse_du: du = dt·W1 inside the row kernel below, no [M, D] round trip
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.wgrad(dtt, u, G["project1.weight"], G["project1.bias"])
            dtt.record_stream(side)
            u.record_stream(side)
        if not fuse_du:
            ops.gemm(dtt, w1t, out_t=du)
        if side is None:
            ops.wgrad(dtt, u, G["project1.weight"], G["project1.bias"])
        need_dx = ctx.needs_input_grad[0]
        # the T copy of dx is the A operand of the preceding block's fc2 data-gradient GEMM (VitBlockFn.backward asks for it with
        # allow_kb): K-blocked when that launch runs on the ring kernels
        dx_t = _act(M, D, dt, x, 4 * D) if (need_dx and dt != torch.float32) else None
        w1t_rows = (dtt, w1t.row if isinstance(w1t, ops.PackedW) else w1t) if fuse_du else None
        if need_dx and g3 is not None and fuse_du and getattr(ctx, "g3_out", False):
            # three-byte in, three-byte out: dx never exists in fp32 — its T copy is the hi plane, one low byte per element beside it
            dlo = torch.empty(M, D, device=x.device, dtype=torch.int8)
            ops.mona_pre_bwd(None, x, g3, P["norm.weight"], P["norm.bias"], P["gamma"], P["gammax"], None, dx_t,
                             G["gamma"], G["gammax"], G["norm.weight"], G["norm.bias"], dt_w1t=w1t_rows, dx_lo=dlo)
            dx = publish_grad3(x.shape, x.device, dx_t, dlo)
        else:
            if need_dx and g3 is not None:                   # the block in front of this adapter does not take tokens: decode once, fp32 from here
                dy = grad3_decode(*g3).view(B, N, D)
            dx = torch.empty_like(x) if need_dx else None
            ops.mona_pre_bwd(du, x, dy if need_dx else None, P["norm.weight"], P["norm.bias"], P["gamma"], P["gammax"], dx, dx_t,
                             G["gamma"], G["gammax"], G["norm.weight"], G["norm.bias"], dt_w1t=w1t_rows)
            if need_dx:
                publish_t_copy(dx, dx_t)
        grads = tuple(None if direct else (G[k] if ctx.needs_input_grad[7 + i] else None) for i, k in enumerate(names))
        return (dx, None, None, None, None, None, None) + grads


_WGRAD_SIDE = {}


def _wgrad_side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _WGRAD_SIDE:
        _WGRAD_SIDE[key] = torch.cuda.Stream(device=device)
    return _WGRAD_SIDE[key]


def set_wgrad_side_stream(flag):
    """Opt-in: MonaFn.backward launches its two uia_wgrad calls on a second HIP stream (join with join_side_streams() before the gradients are read)."""
    _STATE["wgrad_side_stream"] = bool(flag)


def join_side_streams():
    cur = torch.cuda.current_stream()
    for st in _WGRAD_SIDE.values():
        cur.wait_stream(st)


DIRECT_TRAIN = True     # LinearTrainFn / LayerNormAffineFn (fully trainable decoder and heads): weight gradients accumulated straight into the flat-buffer views (A/B: bench.py --no-direct-train-grads)


def _is_flat_grad(p):
    g = p.grad
    return (getattr(p, "_uia_flat_grad", False) and g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape
            and g.device == p.device)


def mona_apply(x_bnd, module_params, variant, hw, p_drop, training, keep_mask=None):
    """module_params: ordered {relative name: Parameter}."""
    names = tuple(k for k in MONA_PARAM_ORDER if k in module_params)
    pd = p_drop if (training or keep_mask is not None) else 0.0
    params = [module_params[k] for k in names]
    direct = torch.is_grad_enabled() and all(p.requires_grad and _is_flat_grad(p) for p in params)
    return MonaFn.apply(x_bnd, variant, tuple(hw), pd, keep_mask, names, direct, *params)


# ================================================================================================ ViT block
class BlockSpec:
    """Static description of a pre-LN block: which parameters play which role."""

    def __init__(self, heads, eps, act, ln1, qkv, proj, ln2, fc1, fc2, mask=None, publish_out=False):
        self.heads, self.eps, self.act, self.mask = heads, eps, act, mask
        self.publish_out = publish_out      # the next consumer of this block's output is another frozen block's LayerNorm (set_ln_fold): leave it T rows + sums
        self.ln1, self.qkv, self.proj, self.ln2, self.fc1, self.fc2 = ln1, qkv, proj, ln2, fc1, fc2   # each: (weight, bias)


class VitBlockFn(torch.autograd.Function):
    """x + attn(LN1 x);  · + mlp(LN2 ·)   with frozen weights: the backward is dgrad only."""

    @staticmethod
    def forward(ctx, x, spec):
        B, N, D = x.shape
        M, dt = B * N, compute_dtype()
        x3 = fwd3_of(x)                                   # the adapter in front handed its output over as a three-byte tensor (set_fwd_resid3): (hi = T rows, low bytes, row sums)
        like = x3[1] if x3 is not None else x
        if x3 is None:
            x = x.contiguous()
            x2d = x.view(M, D)
        train = ctx.needs_input_grad[0]
        fold = ln_fold_enabled(dt, M)
        if x3 is not None:
            assert fold and dt == torch.bfloat16, "a three-byte forward token reached a block that cannot take it"
            rows = (x3[0], x3[2])
        else:
            rows = take_rows(x, dt) if fold else None    # (T copy of x, row sums) left by the GEMM that produced x
        qkv = _empty((M, 3 * D), dt, like)
        if rows is not None:                              # LN1 folded into the QKV GEMM
            wq, cq, bq = WEIGHTS.get_lnfold(spec.qkv[0], spec.qkv[1], spec.ln1[0], spec.ln1[1], dt)
            xin_t = rows[0] if ops.is_kb(rows[0]) else rows[0].view(M, D)
            ops.gemm(xin_t, wq, bias=bq, out_t=qkv, lnfold=(rows[1], cq, D, spec.eps))
            h1 = _act(M, D, dt, like, spec.fc1[0].shape[0]) if x3 is not None else xin_t     # (fp32 x: the T copy's storage is free after this GEMM; three-byte x: it IS x's hi plane and lives on)
        else:
            h1 = _empty((M, D), dt, x)
            ops.layernorm_fwd(x2d, spec.ln1[0], spec.ln1[1], spec.eps, y_t=h1)
            ops.gemm(h1, WEIGHTS.get(spec.qkv[0], dt), bias=spec.qkv[1], out_t=qkv)
        a = _attn_act(M, D, dt, like, D) if D == 64 * spec.heads else _empty((M, D), dt, like)     # read by the output projection (and the backward kernel)
        lse = torch.empty(B, spec.heads, N, device=like.device, dtype=torch.float32) if train else None
        ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, B, spec.heads, N, lse=lse, mask=spec.mask)
        F = spec.fc1[0].shape[0]
        f = _act(M, F, dt, like, D)                          # fc1's result is read by fc2 only
        pre = _empty((M, F), dt, like) if train else None
        # x1 (the attention half's output) never leaves the block: with the fold it can be a THREE-BYTE tensor — the T copy fc1 reads anyway plus one
        # low byte per element (set_block_resid3, bf16, ring tile configs) — instead of fp32 + T copy: the projection's epilogue writes 3 bytes per
        # element instead of 6, fc2's reads 3 instead of 4, and so does the LayerNorm backward that recomputes its statistics (and 39 MB less per block
        # are held for the backward at 256 images)
        # MEASURED, NOT ADOPTED (default off): the two GEMM epilogues gain 0.16 ms per step, the two LayerNorm backwards lose 0.2 — their three planes
        # (2-, 1- and 4-byte elements) cannot all be read with 16-byte accesses per lane, and the row kernel turns from byte- into access-bound
        # (74.7 vs 69.0 us per launch, profiles/r04_c_ab_block_resid3.txt).
        r3 = fold and _STATE.get("block_resid3", False) and dt == torch.bfloat16 and M > 2048
        x1 = None if r3 else torch.empty(M, D, device=like.device, dtype=torch.float32)
        lo1 = torch.empty(M, D, device=like.device, dtype=torch.int8) if r3 else None
        if fold:                                          # LN2 folded: proj leaves T rows + sums, fc1 normalises its accumulators
            sums1 = zero_sums(M, like.device)
            h1 = _as_act(h1, M, D, dt, F) if x3 is None else h1
            resx = dict(resid3=(x3[0], x3[1])) if x3 is not None else dict(resid=x2d)
            if r3:
                ops.gemm(a, WEIGHTS.get(spec.proj[0], dt), bias=spec.proj[1], out_t=h1, out_lo=lo1, rowsum=sums1, **resx)
            else:
                ops.gemm(a, WEIGHTS.get(spec.proj[0], dt), bias=spec.proj[1], out32=x1, out_t=h1, rowsum=sums1, **resx)
            w1, c1, b1 = WEIGHTS.get_lnfold(spec.fc1[0], spec.fc1[1], spec.ln2[0], spec.ln2[1], dt)
            ops.gemm(h1, w1, bias=b1, act=spec.act, aux_out=pre, out_t=f, lnfold=(sums1, c1, D, spec.eps))
        else:
            ops.gemm(a, WEIGHTS.get(spec.proj[0], dt), bias=spec.proj[1], resid=x2d, out32=x1)
            h1 = _as_act(h1, M, D, dt, 0)                                              # row-major: the LayerNorm kernel writes it
            ops.layernorm_fwd(x1, spec.ln2[0], spec.ln2[1], spec.eps, y_t=h1)          # h1 buffer reused as h2
            ops.gemm(h1, WEIGHTS.get(spec.fc1[0], dt), bias=spec.fc1[1], act=spec.act, aux_out=pre, out_t=f)
        x2 = torch.empty(B, N, D, device=like.device, dtype=torch.float32)
        res1 = dict(resid3=(h1, lo1)) if r3 else dict(resid=x1)
        if fold and spec.publish_out:
            sums2 = zero_sums(M, like.device)
            h2 = _act(M, D, dt, like, 3 * D) if r3 else _as_act(h1, M, D, dt, 3 * D)      # (three-byte: h1 is x1's hi plane and stays alive)
            ops.gemm(f, WEIGHTS.get(spec.fc2[0], dt), bias=spec.fc2[1], out32=x2.view(M, D), out_t=h2, rowsum=sums2, **res1)
            publish_rows(x2, h2, sums2)
        else:
            ops.gemm(f, WEIGHTS.get(spec.fc2[0], dt), bias=spec.fc2[1], out32=x2.view(M, D), **res1)
        if train:
            ctx.a_kb = ops.is_kb(a)                       # save_for_backward takes tensors: the K-blocked wrapper is rebuilt in backward
            ctx.r3, ctx.h1_kb = r3, bool(r3 and ops.is_kb(h1))
            ctx.x3_kb = None if x3 is None else bool(ops.is_kb(x3[0]))       # three-byte block input: its two planes are what the LayerNorm backward recomputes from
            ctx.xshape = (B, N, D)
            xs = (x,) if x3 is None else ((x3[0].t if ctx.x3_kb else x3[0]), x3[1])
            if r3:
                ctx.save_for_backward(*xs, qkv, a.t if ctx.a_kb else a, lse, h1.t if ctx.h1_kb else h1, pre, lo1)
            else:
                ctx.save_for_backward(*xs, qkv, a.t if ctx.a_kb else a, lse, x1, pre)
            ctx.spec = spec
            ctx.g3_out = dt == torch.bfloat16 and _g3_partner_feeds(x)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        g3 = grad3_of(dx2)                                            # a three-byte gradient from the adapter behind this block: (T copy, possibly K-blocked; low bytes)
        saved = list(ctx.saved_tensors)
        if ctx.x3_kb is None:
            x = saved.pop(0)
            xin = None
        else:                                                          # three-byte block input (set_fwd_resid3): (hi plane, low bytes)
            xh, xl = saved.pop(0), saved.pop(0)
            xin = (ops.KBlocked(xh) if ctx.x3_kb else xh, xl)
            x = xl                                                     # device / allocation reference below
        if ctx.r3:
            qkv, a, lse, h1, pre, lo1 = saved
            x1 = (ops.KBlocked(h1) if ctx.h1_kb else h1, lo1)         # three-byte x1: (hi plane, low bytes)
        else:
            qkv, a, lse, x1, pre = saved
        if ctx.a_kb:
            a = ops.KBlocked(a)
        spec = ctx.spec
        B, N, D = ctx.xshape
        M, dt = B * N, qkv.dtype
        F = pre.shape[1]
        if g3 is not None:
            dx2_t = g3[0] if ops.is_kb(g3[0]) else g3[0].view(M, D)
            dres2 = (dx2_t, g3[1])
        else:
            dx2 = dx2.contiguous()
            dx2_t = t_copy_of(dx2, dt, allow_kb=True)
            if not ops.is_kb(dx2_t):
                dx2_t = dx2_t.view(M, D)
            dres2 = dx2.view(M, D)
        g3_mode = g3 is not None or getattr(ctx, "g3_out", False)     # three-byte at either end: dx1, which never leaves the block, travels that way too
        # fc2 dgrad fused with act'(pre)
        dpre = _act(M, F, dt, x, D)                       # read by the fc1 dgrad GEMM only
        ops.gemm(dx2_t, WEIGHTS.get(spec.fc2[0], dt, transpose=True), dact=spec.act, aux_in=pre, out_t=dpre)
        dh = _empty((M, D), dt, x)
        ops.gemm(dpre, WEIGHTS.get(spec.fc1[0], dt, transpose=True), out_t=dh)
        del dpre
        if ctx.r3 or g3_mode:                                        # dx1 never leaves the block either: (T copy, low bytes), 3 bytes written instead of 4 + 2
            dx1_t = _empty((M, D), dt, x)
            dlo1 = torch.empty(M, D, device=x.device, dtype=torch.int8)
            ops.layernorm_bwd(dh, x1, spec.ln2[0], spec.eps, dres=dres2, dx_t=dx1_t, dx_lo=dlo1)
            dx1 = (dx1_t, dlo1)
        else:
            dx1 = torch.empty_like(x1)
            dx1_t = _empty((M, D), dt, x) if dt != torch.float32 else dx1
            ops.layernorm_bwd(dh, x1, spec.ln2[0], spec.eps, dres=dres2, dx32=dx1, dx_t=dx1_t if dt != torch.float32 else None)
        da = dh                                                                     # reuse
        ops.gemm(dx1_t, WEIGHTS.get(spec.proj[0], dt, transpose=True), out_t=da)
        dqkv = _attn_act(M, 3 * D, dt, x, D) if D == 64 * spec.heads else _empty((M, 3 * D), dt, x)   # read by the QKV dgrad GEMM only
        if ops.is_kb(dqkv):
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, da, lse, dqkv, None, None, B, spec.heads, N, mask=spec.mask)
        else:
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, da, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, spec.heads, N, mask=spec.mask)
        ops.gemm(dqkv, WEIGHTS.get(spec.qkv[0], dt, transpose=True), out_t=da)       # dh1 into the same buffer
        dx_t = _empty((M, D), dt, x) if dt != torch.float32 else None
        xrows = xin if xin is not None else x.view(M, D)
        if getattr(ctx, "g3_out", False):
            dlo = torch.empty(M, D, device=x.device, dtype=torch.int8)
            ops.layernorm_bwd(da, xrows, spec.ln1[0], spec.eps, dres=dx1, dx_t=dx_t, dx_lo=dlo)
            return publish_grad3((B, N, D), x.device, dx_t, dlo), None
        dx = torch.empty(B, N, D, device=x.device, dtype=torch.float32)
        ops.layernorm_bwd(da, xrows, spec.ln1[0], spec.eps, dres=dx1, dx32=dx.view(M, D), dx_t=dx_t)
        publish_t_copy(dx, dx_t)
        return dx, None


def vit_block(x, spec):
    return VitBlockFn.apply(x, spec)


# ================================================================================================ forward-only pieces
class LnResidual:
    """The fp32 residual entering a post-LN sub-layer, kept as the INPUT of the LayerNorm that produces it: (raw sum, per-row
    (mean, rstd), LayerNorm weight, bias).  uia_gemm's epilogue applies the LayerNorm to the rows it reads (resid_ln_*), so the
    LayerNorm kernel writes only the T operand and 8 bytes of statistics per row instead of a second, fp32 copy of its output
    (201 MB per LayerNorm at 65536 x 768: 24 of them per step in the BERT tower)."""
    __slots__ = ("raw", "stats", "w", "b", "dim", "eps", "hi", "lo")

    def __init__(self, raw, stats, w, b, dim=None, eps=None, hi=None, lo=None):
        self.raw, self.stats, self.w, self.b = raw, stats, w, b
        self.dim, self.eps = dim, eps          # dim set: `stats` holds the row sums (Σ, Σ²) a producing GEMM left (set_ln_fold), not (mean, rstd)
        self.hi, self.lo = hi, lo              # three-byte form (set_text_resid3): raw is None, the sum is its bf16 T copy `hi` + int8 low bytes `lo`

    def gemm_kw(self):
        if self.lo is not None:
            return dict(resid3=(self.hi, self.lo), resid_ln=(self.stats, self.w, self.b, self.dim, self.eps))
        if self.stats is None:
            return dict(resid=self.raw)
        if self.dim is None:
            return dict(resid=self.raw, resid_ln=(self.stats, self.w, self.b))
        return dict(resid=self.raw, resid_ln=(self.stats, self.w, self.b, self.dim, self.eps))


def _post_ln(raw, w, b, eps, x_t, stats_buf=None):
    """LayerNorm of a sub-layer sum: writes the T operand into x_t and returns the fp32 residual for the next sum, deferred
    (raw + statistics) or — set_deferred_text_ln(False) — materialised in place of the sum."""
    if _STATE.get("text_ln_deferred", True):
        stats = stats_buf if stats_buf is not None else torch.empty(raw.shape[0], 2, device=raw.device, dtype=torch.float32)
        ops.layernorm_fwd(raw, w, b, eps, y_t=x_t, stats=stats)
        return LnResidual(raw, stats, w, b)
    ops.layernorm_fwd(raw, w, b, eps, y_t=x_t, y32=raw)
    return LnResidual(raw, None, w, b)


def post_ln_embed(e32, ln, x_t):
    """LayerNorm of the embedding sum: the T operand of the first layer + the fp32 residual."""
    return _post_ln(e32, ln.weight, ln.bias, ln.eps, x_t)


def post_ln_layer(res, x_t, L, B, heads, P, keylen, eps=1e-12, cu_seqlens=None, fold_sums=None, fold_out=False):
    """HF BertLayer (post-LN), frozen: returns the new (deferred fp32 residual, T operand) pair.  P: dict of Parameters with the HF
    names relative to `encoder.layer.{i}.`; q/k/v weights are used as one fused [3D, D] matrix.  res: LnResidual.
    cu_seqlens: rows are PACKED valid tokens (un-padded captions); L is then the longest caption and no mask is needed.
    set_ln_fold (bf16): fold_sums = zeroed fp32 [2, M, 2] lets the two LayerNorms of this layer fold into their neighbouring GEMMs: the
    attention-output LayerNorm always (both neighbours are inside the layer), the output LayerNorm when fold_out says that the next
    layer will take (x_t = T copy of the RAW sum, res.dim set) instead of the normalised operand.  A `res` with .dim set is such an input."""
    kb_in = ops.is_kb(x_t)                            # T copy of a raw sum written K-blocked by the previous layer's last GEMM
    M, D = (x_t.rows, x_t.cols) if kb_in else x_t.shape
    dt = x_t.dtype
    # Post-LN: the residual entering each sub-layer IS the previous LayerNorm's output, so in bf16 mode its T copy (the GEMM
    # operand) could serve as the residual too (saves the residual's fp32 read as well).  Measured at full depth
    # (tools/text_residual_error.py, 12 layers, bf16 vs fp32 mode): the text features' error goes from 6.4e-3 to 1.05e-2, past the
    # 1e-2 bound — so it is OFF unless _STATE["text_resid_t"] is set; the default keeps the fp32 residual exactly (LnResidual).
    t_resid = dt != torch.float32 and _STATE.get("text_resid_t", False)
    fold = fold_sums is not None and ln_fold_enabled(dt, M) and not t_resid
    qkv = _empty((M, 3 * D), dt, x_t)
    if res.dim is not None:                           # x_t holds the raw rows of the previous layer's output sum: its LayerNorm folds in here
        wq, cq, bq = WEIGHTS.get_lnfold(P["_qkv_raw"][0], P["_qkv_raw"][1], res.w, res.b, dt)
        ops.gemm(x_t, wq, bias=bq, out_t=qkv, lnfold=(res.stats, cq, D, res.eps))
    else:
        ops.gemm(x_t, P["_qkv_w"](dt), bias=P["_qkv_b"], out_t=qkv)
    a = _attn_act(M, D, dt, x_t, D) if (D == 64 * heads and cu_seqlens is None) else _empty((M, D), dt, x_t)
    if cu_seqlens is not None:
        ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, B, heads, L, cu_seqlens=cu_seqlens)
    else:
        ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, B, heads, L, mask="keypad", keylen=keylen)
    del qkv
    F = P["intermediate.dense.weight"].shape[0]
    f = _act(M, F, dt, x_t, D)
    # Three-byte sub-layer sums (set_text_resid3, default on with the fold): a sum leaves its GEMM as the bf16 T copy the next GEMM reads anyway
    # plus one low byte per element, and enters the next sum's epilogue in that form — 6 epilogue bytes per element instead of 10 on the 23
    # HBM-bound N = 768 launches of the tower (176 -> 145 us at K = 768, 320 -> 296 at K = 3072); 15 stored mantissa bits.
    r3 = fold and _STATE.get("text_resid3", True) and dt == torch.bfloat16 and M > 2048        # the ring tile configs' epilogue reads and writes the form
    if fold:
        lw_a, lb_a = P["attention.output.LayerNorm.weight"], P["attention.output.LayerNorm.bias"]
        s_a_t = _act(M, D, dt, x_t, F)
        if r3:
            lo_a = torch.empty(M, D, device=x_t.device, dtype=torch.int8)
            ops.gemm(a, WEIGHTS.get(P["attention.output.dense.weight"], dt), bias=P["attention.output.dense.bias"], out_t=s_a_t, out_lo=lo_a,
                     rowsum=fold_sums[0], **res.gemm_kw())
            res_a = LnResidual(None, fold_sums[0], lw_a, lb_a, D, eps, hi=s_a_t, lo=lo_a)
        else:
            s_a = torch.empty(M, D, device=x_t.device, dtype=torch.float32)
            ops.gemm(a, WEIGHTS.get(P["attention.output.dense.weight"], dt), bias=P["attention.output.dense.bias"], out32=s_a, out_t=s_a_t,
                     rowsum=fold_sums[0], **res.gemm_kw())
            res_a = LnResidual(s_a, fold_sums[0], lw_a, lb_a, D, eps)
        w1, c1, b1 = WEIGHTS.get_lnfold(P["intermediate.dense.weight"], P["intermediate.dense.bias"], lw_a, lb_a, dt)
        ops.gemm(s_a_t, w1, bias=b1, act="gelu", out_t=f, lnfold=(fold_sums[0], c1, D, eps))
        if not r3:
            del s_a_t
    else:
        s_a = torch.empty(M, D, device=x_t.device, dtype=torch.float32)
        ops.gemm(a, WEIGHTS.get(P["attention.output.dense.weight"], dt), bias=P["attention.output.dense.bias"], out32=s_a,
                 **(dict(resid_t=x_t) if t_resid else res.gemm_kw()))
        x_t = _as_act(x_t, M, D, dt, 0)
        res_a = _post_ln(s_a, P["attention.output.LayerNorm.weight"], P["attention.output.LayerNorm.bias"], eps, x_t)
        ops.gemm(x_t, WEIGHTS.get(P["intermediate.dense.weight"], dt), bias=P["intermediate.dense.bias"], act="gelu", out_t=f)
    # the previous sub-layer sum was last read by the attention-output GEMM above: its storage takes the new sum
    s_o = res.raw if res.raw is not None else torch.empty(M, D, device=x_t.device, dtype=torch.float32)
    if fold and fold_out:                             # the next layer folds this LayerNorm into its QKV GEMM: x_t becomes the T copy of the raw sum
        x_t = _as_act(x_t, M, D, dt, 3 * D)
        if r3:
            lo_o = res.lo if res.lo is not None else torch.empty(M, D, device=x_t.device, dtype=torch.int8)
            ops.gemm(f, WEIGHTS.get(P["output.dense.weight"], dt), bias=P["output.dense.bias"], out_t=x_t, out_lo=lo_o, rowsum=fold_sums[1],
                     **res_a.gemm_kw())
            return LnResidual(None, fold_sums[1], P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], D, eps, hi=x_t, lo=lo_o), x_t
        ops.gemm(f, WEIGHTS.get(P["output.dense.weight"], dt), bias=P["output.dense.bias"], out32=s_o, out_t=x_t, rowsum=fold_sums[1],
                 **res_a.gemm_kw())
        return LnResidual(s_o, fold_sums[1], P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], D, eps), x_t
    ops.gemm(f, WEIGHTS.get(P["output.dense.weight"], dt), bias=P["output.dense.bias"], out32=s_o,
             **(dict(resid_t=x_t) if t_resid else res_a.gemm_kw()))
    stats_buf = res.stats if (res.dim is None and res.stats is not None) else None
    x_t = _as_act(x_t, M, D, dt, 0)                   # the LayerNorm kernel writes row-major
    return _post_ln(s_o, P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], eps, x_t, stats_buf=stats_buf), x_t


_GRAY_W = {}


def gray_conv_weight(conv_w):
    """Patch-embedding weight for ONE-channel input: Σ_c W[:, c] — a convolution over three identical channels (the reference repeats the grayscale ultrasound
    image, src/datasets/segmentation.py:199-200) is the convolution of the one channel with the summed kernel.  The segmentation loops hand the towers the
    one-channel batch as it arrived from the host: a third of the bytes read, and no 77 MB widened copy written per iteration (round 6: rewriting the input
    batch every iteration costs the step 4 % — it evicts the frozen weights from the 256 MB infinity cache that a resident batch leaves warm).  Cached per
    (tensor, version); frozen weights only."""
    if conv_w.requires_grad:
        raise NotImplementedError("one-channel input through a TRAINABLE patch embedding: widen the batch instead (src.datasets.segmentation.as_model_input)")
    key = id(conv_w)
    hit = _GRAY_W.get(key)
    sig = (conv_w._version, conv_w.data_ptr(), conv_w.dtype, conv_w.device)
    if hit is None or hit[0] != sig or hit[2]() is not conv_w:
        import weakref
        hit = _GRAY_W[key] = (sig, conv_w.detach().float().sum(dim=1, keepdim=True).contiguous(), weakref.ref(conv_w))
    return hit[1]


class PatchEmbedFn(torch.autograd.Function):
    """images [B,3,H,W] → tokens [B, 1+gh*gw, D] fp32 (= cat(cls, conv(x)) + pos); frozen → no backward."""

    @staticmethod
    def forward(ctx, images, conv_w, conv_b, cls, pos, patch):
        dt = compute_dtype()
        B, C, H, W = images.shape
        G = (H // patch) * (W // patch)
        D = conv_w.shape[0]
        K = C * patch * patch
        gran = 64 if dt == torch.bfloat16 else 32                                    # uia_gemm's K granule (128 bytes)
        Kp = (K + gran - 1) // gran * gran                                           # ViT-L/14: 588 → 640, zero-padded on both operands
        cols = _empty((B * G, Kp), dt, images)
        ops.im2col(images.contiguous().float(), cols, patch)
        x = torch.empty(B, G + 1, D, device=images.device, dtype=torch.float32)
        pos2d = pos.detach().reshape(G + 1, D).contiguous()
        ops.gemm(cols, WEIGHTS.get(conv_w, dt, pad_cols_to=Kp), bias=conv_b, resid=pos2d, resid_mod=G, resid_row_off=1, out_group=G, out32=x.view(-1, D))
        ops.fill_cls(x, cls.detach().reshape(D).contiguous(), pos2d[0])
        return x

    @staticmethod
    def backward(ctx, g):
        return None, None, None, None, None, None


class ClsHeadFn(torch.autograd.Function):
    """feat = LN(x[:,0]) @ Wᵀ  (final norm + CLS pool + bias-free projection); frozen weights, dgrad to x."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, w, w_is_in_out=False):
        """w: [out, in] (nn.Linear weight, timm head.proj) or, with w_is_in_out, [in, out] (OpenAI `x @ proj`)."""
        B, N, D = x.shape
        dt = head_dtype()                                   # fp32 operands by default (set_fp32_heads): B rows of work, undamped rounding
        x = x.contiguous()
        h = _empty((B, D), dt, x)
        ops.layernorm_fwd(x, ln_w, ln_b, eps, y_t=h, rows=B, ldx=N * D)
        E = w.shape[1] if w_is_in_out else w.shape[0]
        feat = torch.empty(B, E, device=x.device, dtype=torch.float32)
        ops.gemm(h, WEIGHTS.get(w, dt, transpose=w_is_in_out), out32=feat)
        ctx.save_for_backward(x, ln_w)
        ctx.meta = (eps, w, w_is_in_out, dt)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        x, ln_w = ctx.saved_tensors
        eps, w, w_is_in_out, hdt = ctx.meta
        B, N, D = x.shape
        dt = compute_dtype()
        df = t_copy_of(dfeat.contiguous(), hdt)
        dh = _empty((B, D), hdt, x)
        E = df.shape[1]
        if E % 64:                                          # an embedding narrower than the GEMM's K granule (toy geometries: the reference's own test-size CLIP has 16): zero-padded contraction
            Ep = (E + 63) // 64 * 64
            dfp = df.new_zeros(B, Ep)
            dfp[:, :E] = df
            df = dfp
            wt = WEIGHTS.get(w, hdt, transpose=not w_is_in_out, **({"pad_cols_to": Ep} if w_is_in_out else {"pad_rows_to": Ep}))
        else:
            wt = WEIGHTS.get(w, hdt, transpose=not w_is_in_out)
        ops.gemm(df, wt, out_t=dh)
        if hdt != dt:                                       # the LayerNorm backward writes dx's T copy in the dtype of its dy: hand it a T copy of dh
            dh_t = _empty((B, D), dt, x)
            ops.cast(dh, dh_t)
            dh = dh_t
        dx = torch.zeros_like(x)
        dx_t = torch.zeros(B * N, D, device=x.device, dtype=dt) if dt != torch.float32 else None
        ops.layernorm_bwd(dh, x, ln_w, eps, dx32=dx, dx_t=dx_t, rows=B, ldx=N * D)
        publish_t_copy(dx, dx_t)
        return dx, None, None, None, None, None


# ================================================================================================ op-level functions
# Used where a block cannot take the fully fused frozen path (LoRA on the attention projections).
class LayerNormFn(torch.autograd.Function):
    """fp32 rows → T operand; frozen affine.  backward returns the LN branch of dx only (autograd adds the residual branch)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        y = _empty(x2.shape, compute_dtype(), x)
        ops.layernorm_fwd(x2, w, b, eps, y_t=y)
        ctx.save_for_backward(x2, w)
        ctx.meta = (eps, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        eps, shape = ctx.meta
        dx = torch.empty_like(x2)
        ops.layernorm_bwd(dy.contiguous().view(x2.shape), x2, w, eps, dx32=dx)
        return dx.view(shape), None, None, None


class AttentionFn(torch.autograd.Function):
    """q,k,v: [M, D] T tensors (rows ordered (b, l), row stride shared) → [M, D]."""

    @staticmethod
    def forward(ctx, q, k, v, B, H, L, mask):
        """mask: None | "causal" | ("keypad", keylen int32 [B])."""
        kind, keylen = (mask[0], mask[1]) if isinstance(mask, tuple) else (mask, None)
        out = _empty((B * L, H * 64), q.dtype, q)
        need = any(ctx.needs_input_grad[:3])
        lse = torch.empty(B, H, L, device=q.device, dtype=torch.float32) if need else None
        ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, mask=kind, keylen=keylen)
        if need:
            ctx.save_for_backward(q, k, v, out, lse)
            ctx.meta = (B, H, L, kind, keylen)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        B, H, L, kind, keylen = ctx.meta
        D = H * 64
        dqkv = _empty((B * L, 3 * D), q.dtype, q)
        ops.attn_bwd(q, k, v, out, dout.contiguous(), lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=kind, keylen=keylen)
        return dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], None, None, None, None


class MlpHalfFn(torch.autograd.Function):
    """x + fc2(act(fc1(LN2 x))) with frozen weights (second half of VitBlockFn)."""

    @staticmethod
    def forward(ctx, x1, spec):
        shape = x1.shape
        D = shape[-1]
        x1 = x1.contiguous().view(-1, D)
        M, dt = x1.shape[0], compute_dtype()
        train = ctx.needs_input_grad[0]
        h = _empty((M, D), dt, x1)
        ops.layernorm_fwd(x1, spec.ln2[0], spec.ln2[1], spec.eps, y_t=h)
        F = spec.fc1[0].shape[0]
        f = _empty((M, F), dt, x1)
        pre = _empty((M, F), dt, x1) if train else None
        ops.gemm(h, WEIGHTS.get(spec.fc1[0], dt), bias=spec.fc1[1], act=spec.act, aux_out=pre, out_t=f)
        x2 = torch.empty_like(x1)
        ops.gemm(f, WEIGHTS.get(spec.fc2[0], dt), bias=spec.fc2[1], resid=x1, out32=x2)
        if train:
            ctx.save_for_backward(x1, pre)
            ctx.meta = (spec, shape)
        return x2.view(shape)

    @staticmethod
    def backward(ctx, dx2):
        x1, pre = ctx.saved_tensors
        spec, shape = ctx.meta
        M, D = x1.shape
        dt = pre.dtype
        dx2 = dx2.contiguous().view(M, D)
        dx2_t = t_copy_of(dx2, dt)
        dpre = _empty(pre.shape, dt, x1)
        ops.gemm(dx2_t, WEIGHTS.get(spec.fc2[0], dt, transpose=True), dact=spec.act, aux_in=pre, out_t=dpre)
        dh = _empty((M, D), dt, x1)
        ops.gemm(dpre, WEIGHTS.get(spec.fc1[0], dt, transpose=True), out_t=dh)
        dx1 = torch.empty_like(x1)
        dx1_t = _empty((M, D), dt, x1) if dt != torch.float32 else None
        ops.layernorm_bwd(dh, x1, spec.ln2[0], spec.eps, dres=dx2, dx32=dx1, dx_t=dx1_t)
        publish_t_copy(dx1, dx1_t)                                 # the attention half below starts with a data-gradient GEMM on it
        return dx1.view(shape), None


LORA_PAD = 64     # rank is zero-padded to a multiple of the GEMM's K granule; padded rows/columns contribute exactly 0


def _rank_pad(r):
    """Padded rank: the next multiple of 64 (r = 16 -> 64, r = 96 -> 128).  Every [., rank] buffer of the rank-form path is
    allocated with this width, so no GEMM ever writes N = r columns into a narrower row (an out-of-bounds write for r > 64)."""
    return max(LORA_PAD, (int(r) + LORA_PAD - 1) // LORA_PAD * LORA_PAD)


class LoraLinearFn(torch.autograd.Function):
    """y = x·Wᵀ + b + s·drop(x)·Aᵀ·Bᵀ (+ resid32)  in RANK form (never materialises B·A; reference lora.py:78-90 does).
    x: [M, in] T.  Output: T, or fp32 when an fp32 residual is fused in.  W frozen; A, B (and the bias, reference quirk
    SURVEY Appendix C-4) trainable.

    Launches per call (bf16): forward — frozen GEMM, the N = 64 stream kernel with the input dropout applied to its operand in flight
    (the dropped rows leave as a by-product for dA), the rank GEMM onto the result; backward — frozen dgrad, q = dy·B, s·q·A with the
    dropout's backward in its epilogue, two weight gradients accumulated straight into the factors' .grad when those are views of the
    engine's flat buffer (`direct`).  The factors' padded operand forms come from WEIGHTS (one batched uia_pack_weights per step)."""

    @staticmethod
    def forward(ctx, x, weight, bias, A, Bm, scaling, p_drop, resid32, direct=False):
        dt = x.dtype
        M, K = x.shape
        N = weight.shape[0]
        r = A.shape[0]
        y32 = torch.empty(M, N, device=x.device, dtype=torch.float32) if resid32 is not None else None
        y_t = _empty((M, N), dt, x) if resid32 is None else None
        # K extension (ops.LORA_KEXT): t first, then ONE launch [x | t]·[W | s·B]ᵀ; otherwise the frozen GEMM and a rank-update launch onto its result
        kext = (r > 0 and ops.LORA_KEXT and dt == torch.bfloat16 and _rank_pad(r) == 64 and x.is_cuda and ops.KBLOCK_W and x.is_contiguous() and M >= 256
                and K % 32 == 0 and N % 8 == 0)
        if not kext:
            ops.gemm(x, WEIGHTS.get(weight, dt), bias=bias, resid=resid32, out32=y32, out_t=y_t)
        seed = 0
        xd = x
        t = None
        if r > 0:
            rp = _rank_pad(r)
            t = _empty((M, rp), dt, x)
            a_op = WEIGHTS.get(A, dt, pad_rows_to=rp)
            fuse = p_drop > 0 and dt == torch.bfloat16 and rp == 64 and x.is_cuda and x.is_contiguous() and 64 * (2 * K + 16) <= 160 * 1024 and K % 32 == 0
            regen = fuse and ops.LORA_REGEN_DROP and K % 64 == 0       # the dropped rows are never written: dA's launch regenerates the mask (ops.wgrad(drop=...))
            if p_drop > 0:
                seed = _next_seed()
                if not regen:
                    xd = torch.empty_like(x)
                if not fuse:
                    ops.dropout(x, xd, p_drop, seed)
            if fuse:
                ops.gemm(x, a_op.row, out_t=t, drop=("a", p_drop, seed, None if regen else xd))
            else:
                ops.gemm(xd, a_op, out_t=t)
            if kext:
                ops.gemm(x, WEIGHTS.get_lora_ext((weight,), (Bm,), scaling, dt, rp), bias=bias, resid=resid32, out32=y32, out_t=y_t, a2=(t, 0))
            else:
                bmat = WEIGHTS.get(Bm, dt, pad_cols_to=rp)                          # [N, rp]
                if y32 is not None:
                    ops.gemm(t, bmat, alpha=scaling, resid=y32, out32=y32)
                else:
                    ops.gemm(t, bmat, alpha=scaling, resid_t=y_t, out_t=y_t)
        ctx.save_for_backward(xd, t if t is not None else x.new_empty(0), weight, A, Bm)
        ctx.meta = (scaling, p_drop, seed, r, bias is not None, resid32 is not None)
        ctx.regen = (p_drop, seed) if (r > 0 and regen) else None
        ctx.direct_params = (A, Bm, bias) if direct else None                       # the Parameter objects: .grad is looked up at BACKWARD time
        return y32 if y32 is not None else y_t

    @staticmethod
    def backward(ctx, dy):
        xd, t, weight, A, Bm = ctx.saved_tensors
        scaling, p_drop, seed, r, has_bias, has_resid = ctx.meta
        dt = xd.dtype
        M, K = xd.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        dy_t = t_copy_of(dy, dt) if dy.dtype == torch.float32 else dy
        dx = _empty((M, K), dt, xd)
        ops.gemm(dy_t, WEIGHTS.get(weight, dt, transpose=True), out_t=dx)
        dA = dB = db = None
        pA, pB, pb = ctx.direct_params if ctx.direct_params is not None else (None, None, None)
        want_b = has_bias and ctx.needs_input_grad[2]
        direct = pA is not None and _is_flat_grad(pA) and _is_flat_grad(pB) and (not want_b or (pb is not None and _is_flat_grad(pb)))
        if want_b and not direct:
            db = torch.zeros(N, device=xd.device, dtype=torch.float32)
        gb = (pb.grad if direct else db) if want_b else None                      # Σ dy rides on the matrix cores of the dB launch
        if r > 0:
            rp = _rank_pad(r)
            q = _empty((M, rp), dt, xd)
            ops.gemm(dy_t, WEIGHTS.get(Bm, dt, transpose=True, pad_cols_to=rp), out_t=q)     # q = dy·B   [M, rp]
            at = WEIGHTS.get(A, dt, transpose=True, pad_rows_to=rp)                          # Aᵀ padded: [K, rp]
            if p_drop > 0:
                ops.gemm(q, at, alpha=scaling, resid_t=dx, out_t=dx, drop=("acc", p_drop, seed))
            else:
                ops.gemm(q, at, alpha=scaling, resid_t=dx, out_t=dx)
            if direct:
                ops.wgrad(dy_t, t, pB.grad, dbias=gb, alpha=scaling)
                ops.wgrad(q, xd, pA.grad, alpha=scaling, drop=ctx.regen)
            else:
                dB = torch.zeros(N, r, device=xd.device, dtype=torch.float32)
                ops.wgrad(dy_t, t, dB, dbias=gb, alpha=scaling)
                dA = torch.zeros(r, K, device=xd.device, dtype=torch.float32)
                ops.wgrad(q, xd, dA, alpha=scaling, drop=ctx.regen)
        elif want_b:
            ops.colsum(dy_t, gb)
        return dx, None, db, dA, dB, None, None, (dy if has_resid else None), None


def _lora_down(x, A, p_drop, rp, out=None):
    """t = drop(x)·Aᵀ ([M, rp]) and the dropped rows (for dA): the dropout rides in the N = 64 stream kernel's operand when that kernel takes
    the shape (bf16, rank padded to 64), otherwise as a pass of its own.  Returns (t, xd, seed)."""
    dt = x.dtype
    M, K = x.shape
    t = _empty((M, rp), dt, x) if out is None else out
    a_op = WEIGHTS.get(A, dt, pad_rows_to=rp)
    if p_drop <= 0:
        ops.gemm(x, a_op, out_t=t)
        return t, x, 0
    seed = _next_seed()
    if dt == torch.bfloat16 and rp == 64 and x.is_cuda and x.is_contiguous() and 64 * (2 * K + 16) <= 160 * 1024 and K % 32 == 0:
        if ops.LORA_REGEN_DROP and K % 64 == 0:
            # the dropped rows are never written: dA's launch regenerates the mask from the seed while it stages x (ops.wgrad(drop=...));
            # callers get x itself back and keep the seed
            ops.gemm(x, a_op.row, out_t=t, drop=("a", p_drop, seed, None))
            return t, x, seed
        xd = torch.empty_like(x)
        ops.gemm(x, a_op.row, out_t=t, drop=("a", p_drop, seed, xd))
    else:
        xd = torch.empty_like(x)
        ops.dropout(x, xd, p_drop, seed)
        ops.gemm(xd, a_op, out_t=t)
    return t, xd, seed


def _lora_grads(dy_t, t, xd, q, A, Bm, bias, scaling, direct, regen=None):
    """dB = s·dyᵀ·t, dA = s·qᵀ·x̃ and, for a bias that trains (reference quirk C-4), db = Σ dy on the matrix cores of the dB launch: into the
    parameters' .grad (flat-buffer views) when `direct`, else returned."""
    r = A.shape[0]
    want_b = bias is not None and bias.requires_grad
    # regen = (p, seed): `xd` is the UN-dropped input and the dA launch regenerates the forward's mask (ops.LORA_REGEN_DROP)
    if direct:
        ops.wgrad(dy_t, t, Bm.grad, dbias=bias.grad if want_b else None, alpha=scaling)
        ops.wgrad(q, xd, A.grad, alpha=scaling, drop=regen)
        return None, None, None
    db = torch.zeros(bias.numel(), device=xd.device, dtype=torch.float32) if want_b else None
    dB = torch.zeros(Bm.shape[0], r, device=xd.device, dtype=torch.float32)
    ops.wgrad(dy_t, t, dB, dbias=db, alpha=scaling)
    dA = torch.zeros(r, A.shape[1], device=xd.device, dtype=torch.float32)
    ops.wgrad(q, xd, dA, alpha=scaling, drop=regen)
    return dA, dB, db


class LoraAttnHalfFn(torch.autograd.Function):
    """x1 = x + proj(attention(q, k, v)(LN1 x)) with LinearLoRA on q, k, v and the output projection (reference lora.py:115-199 inside an
    OpenAI-CLIP residual block, model.py:195-201) as ONE autograd node.  Same arithmetic as LayerNormFn -> 3 x LoraLinearFn -> AttentionFn ->
    LoraLinearFn; what changes is the schedule:
      * q, k, v share ONE frozen GEMM (N = 3D) and ONE data-gradient GEMM (K = 3D) — a third of the launches and M tails, and the three
        gradients of h are summed inside the launches instead of by two autograd adds;
      * the attention backward writes its fused [M, 3D] gradient straight into that GEMM's operand;
      * the LayerNorm backward adds the residual gradient and publishes the T copy for the block below.
    Parameters: W* frozen; b*, A*, B* trainable (bias: reference quirk SURVEY C-4)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, heads, mask, scaling, p_drop, direct, wq, bq, aq, Bq, wk, bk, ak, Bk, wv, bv, av, Bv, wo, bo, ao, Bo):
        shape = x.shape
        D = shape[-1]
        Bsz, L = shape[0], shape[1]
        x2 = x.contiguous().view(-1, D)
        M, dt = x2.shape[0], compute_dtype()
        r = aq.shape[0]
        rp = _rank_pad(r)
        h = _empty((M, D), dt, x2)
        # K extension (bf16, rank padded to 64, ring kernels): the rank update s·t·Bᵀ rides in the frozen GEMM's K loop as 64 more columns of
        # [x | t]·[W | s·B]ᵀ — no read-modify-write pass over the result (lora.py:87; ops.LORA_KEXT)
        kext = ops.LORA_KEXT and dt == torch.bfloat16 and rp == 64 and x.is_cuda and ops.KBLOCK_W and D % 256 == 0 and M >= 256
        t_all = _empty((3, M, rp), dt, x2) if kext else None
        if (kext and x2.is_contiguous() and ops.ln_lora_down_ok(D, r, dt) and ak.shape[0] == r and av.shape[0] == r and (p_drop == 0 or ops.LORA_REGEN_DROP)):
            # LayerNorm and the three down-projections in ONE launch: the h tile goes through LDS to the matrix cores, h is written once and never read back
            seeds3 = [_next_seed() if p_drop > 0 else 0 for _ in range(3)]
            a_rows = [WEIGHTS.get(A, dt, pad_rows_to=rp) for A in (aq, ak, av)]
            ops.ln_lora_down(x2, ln_w, ln_b, eps, h, [a.row if isinstance(a, ops.PackedW) else a for a in a_rows], t_all, p_drop, seeds3)
            downs = [(t_all[i], h, seeds3[i]) for i in range(3)]
        else:
            ops.layernorm_fwd(x2, ln_w, ln_b, eps, y_t=h)
            downs = [_lora_down(h, A, p_drop, rp, out=None if t_all is None else t_all[i]) for i, A in enumerate((aq, ak, av))]
        qkv = _empty((M, 3 * D), dt, x2)
        bcat = torch.cat([b.detach() for b in (bq, bk, bv)]) if bq is not None else None
        if kext:
            ops.gemm(h, WEIGHTS.get_lora_ext((wq, wk, wv), (Bq, Bk, Bv), scaling, dt, rp), bias=bcat, out_t=qkv, a2=(t_all, D))
        else:
            ops.gemm(h, WEIGHTS.get_cat((wq, wk, wv), dt), bias=bcat, out_t=qkv)
            for i, (Bm, (t, _, _)) in enumerate(zip((Bq, Bk, Bv), downs)):
                sl = qkv[:, i * D:(i + 1) * D]
                ops.gemm(t, WEIGHTS.get(Bm, dt, pad_cols_to=rp), alpha=scaling, resid_t=sl, out_t=sl)
        a = _empty((M, D), dt, x2)
        lse = torch.empty(Bsz, heads, L, device=x.device, dtype=torch.float32)
        ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, Bsz, heads, L, lse=lse, mask=mask)
        to, ado, seed_o = _lora_down(a, ao, p_drop, rp)
        x1 = torch.empty_like(x2)
        if kext:
            ops.gemm(a, WEIGHTS.get_lora_ext((wo,), (Bo,), scaling, dt, rp), bias=bo, resid=x2, out32=x1, a2=(to, 0))
        else:
            ops.gemm(a, WEIGHTS.get(wo, dt), bias=bo, resid=x2, out32=x1)
            ops.gemm(to, WEIGHTS.get(Bo, dt, pad_cols_to=rp), alpha=scaling, resid=x1, out32=x1)
        ctx.save_for_backward(x2, ln_w, qkv, a, lse, to, ado, *[d[0] for d in downs], *[d[1] for d in downs], wq, wk, wv, wo)
        ctx.meta = (eps, shape, heads, mask, scaling, p_drop, [d[2] for d in downs] + [seed_o], rp)
        # which of the four saved LoRA inputs are the UN-dropped tensors (h three times, a once): their dA launches regenerate the mask
        ctx.regen = [p_drop > 0 and d[1] is h for d in downs] + [p_drop > 0 and ado is a]
        ctx.params = (bq, aq, Bq, bk, ak, Bk, bv, av, Bv, bo, ao, Bo)                # the Parameter objects: .grad is looked up at BACKWARD time
        ctx.direct = direct
        return x1.view(shape)

    @staticmethod
    def backward(ctx, dx1):
        x2, ln_w, qkv, a, lse, to, ado, tq, tk, tv, hq, hk, hv, wq, wk, wv, wo = ctx.saved_tensors
        eps, shape, heads, mask, scaling, p_drop, seeds, rp = ctx.meta
        bq, aq, Bq, bk, ak, Bk, bv, av, Bv, bo, ao, Bo = ctx.params
        M, D = x2.shape
        Bsz, L = shape[0], shape[1]
        dt = qkv.dtype
        trainable = [p for p in ctx.params if p is not None and p.requires_grad]
        direct = ctx.direct and all(_is_flat_grad(p) for p in trainable)
        dx1 = dx1.contiguous().view(M, D)
        dy_t = t_copy_of(dx1, dt)
        drop = lambda i: ("acc", p_drop, seeds[i]) if p_drop > 0 else None
        # ---- output projection
        da = _empty((M, D), dt, x2)
        ops.gemm(dy_t, WEIGHTS.get(wo, dt, transpose=True), out_t=da)
        qo = _empty((M, rp), dt, x2)
        ops.gemm(dy_t, WEIGHTS.get(Bo, dt, transpose=True, pad_cols_to=rp), out_t=qo)
        ops.gemm(qo, WEIGHTS.get(ao, dt, transpose=True, pad_rows_to=rp), alpha=scaling, resid_t=da, out_t=da, drop=drop(3))
        g_o = _lora_grads(dy_t, to, ado, qo, ao, Bo, bo, scaling, direct, regen=(p_drop, seeds[3]) if ctx.regen[3] else None)
        # ---- attention
        dqkv = _empty((M, 3 * D), dt, x2)
        ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, da, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], Bsz, heads, L, mask=mask)
        # ---- q, k, v  (the first block of a tower with a frozen embedding has nobody to hand dx to: no data-gradient GEMM, no LayerNorm backward)
        need_dx = ctx.needs_input_grad[0]
        dh = _empty((M, D), dt, x2) if need_dx else None
        if need_dx:
            ops.gemm(dqkv, WEIGHTS.get_cat((wq, wk, wv), dt, transpose=True), out_t=dh)
        grads = []
        # the three rank terms Σ_i mask_i ⊙ (s·q_i·A_i) reach dh in ONE read-modify-write pass (uia_lora_rank_update) instead of one per projection
        one_pass = need_dx and rp == 64 and x2.is_cuda and ops.LORA_RANK3 and ops.lora_rank_update_ok(3, D, dt)
        q_all = _empty((3, M, rp), dt, x2)
        trio = ((bq, aq, Bq, tq, hq), (bk, ak, Bk, tk, hk), (bv, av, Bv, tv, hv))
        # the three dB launches and the three dA launches as ONE launch each (uia_wgrad_group): same shapes, same strides, per-problem pointers and dropout seeds
        grouped = (direct and ops.LORA_WGRAD_GROUP and x2.is_cuda and dt == torch.bfloat16 and rp == 64 and len({bool(r_) for r_ in ctx.regen[:3]}) == 1
                   and all(t_.shape == tq.shape and t_.stride() == tq.stride() for t_ in (tk, tv)) and all(h_.shape == hq.shape and h_.stride() == hq.stride() for h_ in (hk, hv))
                   and all(P_.shape == Bq.shape for P_ in (Bk, Bv)) and all(P_.shape == aq.shape for P_ in (ak, av))
                   and len({(b_ is not None and b_.requires_grad) for b_ in (bq, bk, bv)}) == 1)
        for i, (bias, A, Bm, t, hd) in enumerate(trio):
            dsl = dqkv[:, i * D:(i + 1) * D]
            qi = q_all[i]
            ops.gemm(dsl, WEIGHTS.get(Bm, dt, transpose=True, pad_cols_to=rp), out_t=qi)
            if need_dx and not one_pass:
                ops.gemm(qi, WEIGHTS.get(A, dt, transpose=True, pad_rows_to=rp), alpha=scaling, resid_t=dh, out_t=dh, drop=drop(i))
            if not grouped:
                grads.append(_lora_grads(dsl, t, hd, qi, A, Bm, bias, scaling, direct, regen=(p_drop, seeds[i]) if ctx.regen[i] else None))
        if grouped:
            want_b = bq is not None and bq.requires_grad
            ops.wgrad_group([dqkv[:, i * D:(i + 1) * D] for i in range(3)], [tq, tk, tv], [Bq.grad, Bk.grad, Bv.grad],
                            dbias_list=[bq.grad, bk.grad, bv.grad] if want_b else None, alpha=scaling)
            ops.wgrad_group([q_all[i] for i in range(3)], [hq, hk, hv], [aq.grad, ak.grad, av.grad], alpha=scaling,
                            drop=(p_drop, seeds[:3]) if ctx.regen[0] else None)
            grads += [(None, None, None)] * 3
        if one_pass:
            ats = [WEIGHTS.get(A, dt, transpose=True, pad_rows_to=rp) for A in (aq, ak, av)]
            ops.lora_rank_update(q_all, [a.row if isinstance(a, ops.PackedW) else a for a in ats], dh, scaling, p_drop, seeds[:3])
        grads.append(g_o)
        # ---- LayerNorm: dx = dx1 + LN'(dh); the T copy goes to the block below (its MLP half starts with a data-gradient GEMM)
        dx = None
        if need_dx:
            dx = torch.empty_like(x2)
            dx_t = _empty((M, D), dt, x2) if dt != torch.float32 else None
            ops.layernorm_bwd(dh, x2, ln_w, eps, dres=dx1, dx32=dx, dx_t=dx_t)
            publish_t_copy(dx, dx_t)
            dx = dx.view(shape)
        flat = []
        for dA, dB, db in grads:                                                    # order of the inputs: (w, b, A, B) per projection
            flat += [None, db, dA, dB]
        return (dx, None, None, None, None, None, None, None, None, *flat)


def _pad_cols(p, dt, transpose=False, rows=False):
    """T copy of a LoRA factor zero-padded to _rank_pad(r) along its rank dimension.
    B [out, r]  -> [out, rp] (transpose=False)  or  Bᵀ -> [rp, out] (transpose=True)
    A [r, in]   -> Aᵀ padded [in, rp] (transpose=True, rows=True)"""
    src = p.detach().float()
    if rows:                                   # A: rank is the row dim
        rp = _rank_pad(src.shape[0])
        if src.shape[0] < rp:
            src = torch.cat([src, src.new_zeros(rp - src.shape[0], src.shape[1])], 0)
        out = torch.empty(src.shape[1], rp, device=src.device, dtype=dt)
        ops.transpose_cast(src.contiguous(), out)
        return out
    rp = _rank_pad(src.shape[1])
    if src.shape[1] < rp:                      # B: rank is the column dim
        src = torch.cat([src, src.new_zeros(src.shape[0], rp - src.shape[1])], 1)
    src = src.contiguous()
    if transpose:
        out = torch.empty(rp, src.shape[0], device=src.device, dtype=dt)
        ops.transpose_cast(src, out)
        return out
    if dt == torch.float32:
        return src
    out = torch.empty(src.shape, device=src.device, dtype=dt)
    ops.cast(src, out)
    return out


# ================================================================================================ trainable op-level functions
# (CLIPSeg decoder: every weight trains, so these produce dx, dW and db.  wgrad needs both GEMM dimensions to be
#  multiples of 64: callers zero-pad narrow outputs, e.g. the last transposed convolution's 16 columns.)
class CastFn(torch.autograd.Function):
    """fp32 → compute dtype (operand copy); the backward is the plain upcast."""

    @staticmethod
    def forward(ctx, x):
        dt = compute_dtype()
        if dt == torch.float32:
            return x.contiguous()
        out = torch.empty(x.shape, device=x.device, dtype=dt)
        ops.cast(x.contiguous(), out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g.float()


class LinearTrainFn(torch.autograd.Function):
    """y = act(x·Wᵀ + b) (+ resid32), x [M,K] in T, W [N,K] / b [N] fp32 trainable.  Output T, or fp32 when `out32` / a residual."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, resid32, out32):
        dt = x.dtype
        M, K = x.shape
        N = weight.shape[0]
        want32 = out32 or resid32 is not None
        assert not (act and want32), "activation + fp32 output is not used by any caller"
        y = torch.empty(M, N, device=x.device, dtype=torch.float32 if want32 else dt)
        w_t = WEIGHTS.get(weight, dt)
        stash = x.new_empty(0)
        if want32:
            ops.gemm(x, w_t, bias=bias, resid=resid32, out32=y)
        elif act in ("gelu", "quick_gelu"):                     # smooth activations need the pre-activation in the backward
            stash = torch.empty_like(y)
            ops.gemm(x, w_t, bias=bias, act=act, aux_out=stash, out_t=y)
        else:
            ops.gemm(x, w_t, bias=bias, act=act, out_t=y)
            if act:
                stash = y                                       # ReLU: the post-activation is enough
        ctx.save_for_backward(x, weight, stash)
        ctx.meta = (act, bias is not None, resid32 is not None)
        # the Parameter objects (when the operands ARE parameters, not views or concatenations of them): their .grad is looked up at BACKWARD time, and
        # when it is a view of the engine's flat gradient buffer the weight gradient is accumulated straight into it (no zero fill, no AccumulateGrad add)
        ctx.params = (weight if isinstance(weight, torch.nn.Parameter) else None, bias if isinstance(bias, torch.nn.Parameter) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        act, has_bias, has_resid = ctx.meta
        dt = x.dtype
        M, K = x.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        dy_t = t_copy_of(dy, dt) if dy.dtype == torch.float32 else dy
        if act:
            dpre = torch.empty_like(dy_t)
            ops.act_bwd(dy_t, y, act, dpre)
            dy_t = dpre
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty((M, K), dt, x)
            ops.gemm(dy_t, WEIGHTS.get(weight, dt, transpose=True), out_t=dx)
        pw, pb = ctx.params
        if (DIRECT_TRAIN and pw is not None and _is_flat_grad(pw) and ctx.needs_input_grad[1] and
                (not has_bias or (pb is not None and _is_flat_grad(pb) and ctx.needs_input_grad[2]))):
            ops.wgrad(dy_t, x, pw.grad, pb.grad if has_bias else None)
            return dx, None, None, None, (dy if has_resid else None), None
        dW = torch.zeros(N, K, device=x.device, dtype=torch.float32)
        db = torch.zeros(N, device=x.device, dtype=torch.float32) if has_bias else None
        ops.wgrad(dy_t, x, dW, db)
        return dx, dW, db, None, (dy if has_resid else None), None


class LayerNormAffineFn(torch.autograd.Function):
    """fp32 rows → fp32 rows, trainable γ/β (CLIPSeg decoder layer_norm1/2)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        y = torch.empty_like(x2)
        ops.layernorm_fwd(x2, w, b, eps, y32=y)
        ctx.save_for_backward(x2, w)
        ctx.meta = (eps, x.shape)
        ctx.params = (w if isinstance(w, torch.nn.Parameter) else None, b if isinstance(b, torch.nn.Parameter) else None)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        eps, shape = ctx.meta
        dx = torch.empty_like(x2)
        pw, pb = ctx.params
        if DIRECT_TRAIN and pw is not None and pb is not None and _is_flat_grad(pw) and _is_flat_grad(pb) and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            ops.layernorm_bwd_affine(dy.contiguous().view(x2.shape).float(), x2, w, eps, dx, pw.grad, pb.grad)      # the kernel ADDS its per-block sums
            return dx.view(shape), None, None, None
        gw, gb = torch.zeros_like(w, dtype=torch.float32), torch.zeros_like(w, dtype=torch.float32)
        ops.layernorm_bwd_affine(dy.contiguous().view(x2.shape).float(), x2, w, eps, dx, gw, gb)
        return dx.view(shape), gw, gb, None


class SmallAttentionFn(torch.autograd.Function):
    """q,k,v [M, H·dh] (strided views of one fused buffer), dh ∈ {16, 32, 64} → [M, H·dh]."""

    @staticmethod
    def forward(ctx, q, k, v, B, H, L, dh):
        out = _empty((B * L, H * dh), q.dtype, q)
        lse = torch.empty(B, H, L, device=q.device, dtype=torch.float32)
        ops.attn_fwd(q, k, v, out, B, H, L, lse=lse, dh=dh)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta = (B, H, L, dh)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        B, H, L, dh = ctx.meta
        D = H * dh
        dqkv = _empty((B * L, 3 * D), q.dtype, q)
        ops.attn_bwd(q, k, v, out, dout.contiguous(), lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, dh=dh)
        return dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], None, None, None, None


class FilmFn(torch.autograd.Function):
    """y[b,n,:] = mul[b,:]·x[b,n,:] + add[b,:]   (fp32)."""

    @staticmethod
    def forward(ctx, x, mul, add):
        x, mul, add = x.contiguous(), mul.contiguous(), add.contiguous()
        y = torch.empty_like(x)
        ops.film_fwd(x, mul, add, y)
        ctx.save_for_backward(x, mul)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mul = ctx.saved_tensors
        dx, dmul, dadd = torch.empty_like(x), torch.empty_like(mul), torch.empty_like(mul)
        ops.film_bwd(dy.contiguous(), x, mul, dx, dmul, dadd)
        return dx, dmul, dadd


class Im2col3x3Fn(torch.autograd.Function):
    """tokens [B, 1+h·w, C] fp32 → 3×3 zero-padded patches [B·h·w, 9C] in T (CLS token dropped, as decoder.forward does)."""

    @staticmethod
    def forward(ctx, x, h, w):
        B, N, C = x.shape
        cols = _empty((B * h * w, 9 * C), compute_dtype(), x)
        ops.im2col3x3(x.contiguous(), cols, h, w, tok_off=N - h * w)
        ctx.meta = (x.shape, h, w)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        shape, h, w = ctx.meta
        dx = torch.empty(shape, device=dcols.device, dtype=torch.float32)
        ops.col2im3x3(dcols.contiguous(), dx, h, w, tok_off=shape[1] - h * w)
        return dx, None, None


class UnshuffleFn(torch.autograd.Function):
    """[B·h·w·k1², ≥k2²] (output of the two stacked kernel=stride transposed convolutions) → logits [B, h·k1·k2, w·k1·k2] fp32."""

    @staticmethod
    def forward(ctx, tmp, B, h, w, k1, k2):
        out = torch.empty(B, h * k1 * k2, w * k1 * k2, device=tmp.device, dtype=torch.float32)
        ops.unshuffle(tmp, out, B, h, w, k1, k2)
        ctx.meta = (tmp.shape, tmp.dtype, B, h, w, k1, k2)
        return out

    @staticmethod
    def backward(ctx, dout):
        shape, dt, B, h, w, k1, k2 = ctx.meta
        dtmp = torch.empty(shape, device=dout.device, dtype=dt)
        ops.shuffle(dout.contiguous().float(), dtmp, B, h, w, k1, k2)
        return dtmp, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ FPN task heads
class UpsampleBilinearFn(torch.autograd.Function):
    """token-major fp32 [B*h*w, C] → [B, C, H, W] (nn.Upsample(size, mode="bilinear", align_corners=False))."""

    @staticmethod
    def forward(ctx, tok, B, h, w, H, W):
        tok = tok.contiguous()
        C = tok.shape[1]
        out = torch.empty(B, C, H, W, device=tok.device, dtype=torch.float32)
        ops.upsample_bilinear(tok, B, C, h, w, H, W, out)
        ctx.meta = (B, C, h, w, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, h, w, H, W = ctx.meta
        dtok = torch.empty(B * h * w, C, device=dout.device, dtype=torch.float32)
        ops.upsample_bilinear(dout.contiguous().float(), B, C, h, w, H, W, dtok, backward=True)
        return dtok, None, None, None, None, None


class SegmentMeanFn(torch.autograd.Function):
    """fp32 [B*n, C] → [B, C]: mean over each image's n tokens (AdaptiveAvgPool2d(1) + Flatten)."""

    @staticmethod
    def forward(ctx, x, B, n):
        x = x.contiguous()
        out = torch.empty(B, x.shape[1], device=x.device, dtype=torch.float32)
        ops.segment_mean(x, B, n, out)
        ctx.meta = (B, n, x.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        B, n, C = ctx.meta
        dx = torch.empty(B * n, C, device=dout.device, dtype=torch.float32)
        ops.segment_mean(dout.contiguous().float(), B, n, dx, backward=True)
        return dx, None, None


class DropoutFn(torch.autograd.Function):
    """Inverted dropout with the library's counter-hash mask (regenerated, not stored, in the backward)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        y = torch.empty_like(x)
        ops.dropout(x, y, p, seed)
        ctx.meta = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ops.dropout(dy, dx, p, seed)
        return dx, None, None


# ------------------------------------------------------------------------------------------------ trainable post-LN (BERT) path
class FrozenLinearFn(torch.autograd.Function):
    """y = act(x·Wᵀ + b) (+ resid32) with FROZEN W, b: forward like LinearTrainFn, backward is the data gradient only."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, resid32):
        dt = x.dtype
        M, K = x.shape
        N = weight.shape[0]
        need = ctx.needs_input_grad[0] or (resid32 is not None and ctx.needs_input_grad[4])
        if resid32 is not None:
            assert not act
            y = torch.empty(M, N, device=x.device, dtype=torch.float32)
            ops.gemm(x, WEIGHTS.get(weight, dt), bias=bias, resid=resid32, out32=y)
            pre = None
        else:
            y = _empty((M, N), dt, x)
            pre = _empty((M, N), dt, x) if (act and need) else None
            ops.gemm(x, WEIGHTS.get(weight, dt), bias=bias, act=act, aux_out=pre, out_t=y)
        ctx.save_for_backward(weight, pre if pre is not None else x.new_empty(0))
        ctx.meta = (act, resid32 is not None, dt, K)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, pre = ctx.saved_tensors
        act, has_resid, dt, K = ctx.meta
        dy = dy.contiguous()
        dy_t = t_copy_of(dy, dt) if dy.dtype == torch.float32 else dy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty((dy.shape[0], K), dt, dy)
            if act:
                dpre = torch.empty_like(dy_t)
                ops.act_bwd(dy_t, pre, act, dpre)
                ops.gemm(dpre, WEIGHTS.get(weight, dt, transpose=True), out_t=dx)
            else:
                ops.gemm(dy_t, WEIGHTS.get(weight, dt, transpose=True), out_t=dx)
        return dx, None, None, None, (dy if has_resid else None)


class PostLayerNormFn(torch.autograd.Function):
    """Post-LN sub-layer output: LayerNorm(s) with frozen affine, returned both as the fp32 residual and as the T operand.
    Backward: LN backward of the SUM of the two incoming gradients, in fp32."""

    @staticmethod
    def forward(ctx, s32, w, b, eps):
        s32 = s32.contiguous()
        y32 = torch.empty_like(s32)
        y_t = _empty(s32.shape, compute_dtype(), s32) if compute_dtype() != torch.float32 else None
        ops.layernorm_fwd(s32, w, b, eps, y_t=y_t, y32=y32)
        ctx.save_for_backward(s32, w)
        ctx.eps = eps
        return y32, (y_t if y_t is not None else y32)

    @staticmethod
    def backward(ctx, g32, g_t):
        s32, w = ctx.saved_tensors
        g = None
        for t in (g32, g_t):
            if t is not None:
                g = t.float() if g is None else g + t.float()
        ds = torch.empty_like(s32)
        ops.layernorm_bwd(g.contiguous(), s32, w, ctx.eps, dx32=ds)
        return ds, None, None, None


class EmbedFn(torch.autograd.Function):
    """BERT embedding sum word[ids] + position[:L] + token_type[0] → fp32 rows [B·L, D], with gradients for the three tables
    (nn.Embedding padding_idx semantics for the word table)."""

    @staticmethod
    def forward(ctx, ids, word, pos, typ, pad_id):
        B, L = ids.shape
        D = word.shape[1]
        x = torch.empty(B * L, D, device=ids.device, dtype=torch.float32)
        ops.embed(ids, word.detach(), pos.detach(), typ.detach()[0].contiguous(), x)
        ctx.save_for_backward(ids)
        ctx.meta = (word.shape, pos.shape, typ.shape, pad_id, B, L, D)
        return x

    @staticmethod
    def backward(ctx, dx):
        (ids,) = ctx.saved_tensors
        wshape, pshape, tshape, pad_id, B, L, D = ctx.meta
        dx = dx.contiguous()
        dword = torch.zeros(wshape, device=dx.device, dtype=torch.float32)
        ops.embed_bwd(ids, dx, dword, pad_id)
        dpos = torch.zeros(pshape, device=dx.device, dtype=torch.float32)
        ops.colsum(dx.view(B, L * D), dpos.view(-1)[:L * D])             # Σ over the batch per position
        dtyp = torch.zeros(tshape, device=dx.device, dtype=torch.float32)
        ops.colsum(dx, dtyp[0])                                            # every token has type 0
        return None, dword, dpos, dtyp, None


def _train_lin(mod, rows, resid32=None, act=None):
    """One Linear of the trainable text path: LoRA module, trainable plain Linear, or frozen."""
    from src.adapters.lora import LinearLoRA
    if isinstance(mod, LinearLoRA):
        assert act is None
        return mod.apply_rows(rows, resid32)
    if mod.weight.requires_grad or (mod.bias is not None and mod.bias.requires_grad):
        return LinearTrainFn.apply(rows, mod.weight, mod.bias, act, resid32, False)
    return FrozenLinearFn.apply(rows, mod.weight, mod.bias, act, resid32)


def _train_post_ln(s32, ln, eps):
    """LayerNorm of a post-LN sub-layer → (fp32 residual, T operand); trainable affine when it requires grad."""
    if ln.weight.requires_grad or ln.bias.requires_grad:
        y32 = LayerNormAffineFn.apply(s32, ln.weight, ln.bias, eps)
        return y32, CastFn.apply(y32)
    return PostLayerNormFn.apply(s32, ln.weight, ln.bias, eps)


def post_ln_layer_train(x32, x_t, L, B, heads, layer, keylen, eps=1e-12):
    """HF BertLayer (post-LN) with autograd: q/k/v/attention.output.dense may be LinearLoRA modules (reference lora.py:317-367,
    --tune_text_encoder) and any Linear / LayerNorm may itself be trainable (--method full --tune_text_encoder); the rest is
    frozen.  Returns the new (fp32, T) residual pair."""
    sa, ao = layer.attention.self, layer.attention.output
    q, k, v = _train_lin(sa.query, x_t), _train_lin(sa.key, x_t), _train_lin(sa.value, x_t)
    a = AttentionFn.apply(q, k, v, B, heads, L, ("keypad", keylen))
    s = _train_lin(ao.dense, a, resid32=x32)
    x32, x_t = _train_post_ln(s, ao.LayerNorm, eps)
    f = _train_lin(layer.intermediate.dense, x_t, act="gelu")
    s = _train_lin(layer.output.dense, f, resid32=x32)
    return _train_post_ln(s, layer.output.LayerNorm, eps)


# ------------------------------------------------------------------------------------------------ full fine-tuning of the image tower
# (reference biomedclip/finetune.py:134-157, --method full — its argparse default).  Frozen-backbone blocks keep the fused
# VitBlockFn; a block with trainable weights is the same arithmetic composed of the trainable op-level functions
# (LayerNormAffineFn, LinearTrainFn with uia_wgrad, attention): correct first, not tuned — the headline path is the adapters.
class AttentionQkvFn(torch.autograd.Function):
    """Fused-projection attention: qkv [M, 3D] T → [M, D]; the backward writes dq | dk | dv into one [M, 3D] tensor."""

    @staticmethod
    def forward(ctx, qkv, B, H, L, mask):
        D = H * 64
        out = _empty((B * L, D), qkv.dtype, qkv)
        need = ctx.needs_input_grad[0]
        lse = torch.empty(B, H, L, device=qkv.device, dtype=torch.float32) if need else None
        ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, B, H, L, lse=lse, mask=mask)
        if need:
            ctx.save_for_backward(qkv, out, lse)
            ctx.meta = (B, H, L, mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, H, L, mask = ctx.meta
        D = H * 64
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], out, dout.contiguous(), lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, mask=mask)
        return dqkv, None, None, None, None


def trainable_vit_block(x, heads, eps, act, ln1, qkv, proj, ln2, fc1, fc2, mask=None):
    """x [B, N, D] fp32 → [B, N, D]; every (weight, bias) pair may require grad."""
    B, N, D = x.shape
    M = B * N
    xr = x.contiguous().view(M, D)
    h = LayerNormAffineFn.apply(xr, ln1[0], ln1[1], eps)
    q = LinearTrainFn.apply(CastFn.apply(h), qkv[0], qkv[1], None, None, False)
    a = AttentionQkvFn.apply(q, B, heads, N, mask)
    x1 = LinearTrainFn.apply(a, proj[0], proj[1], None, xr, True)
    h2 = LayerNormAffineFn.apply(x1, ln2[0], ln2[1], eps)
    f = LinearTrainFn.apply(CastFn.apply(h2), fc1[0], fc1[1], act, None, False)
    x2 = LinearTrainFn.apply(f, fc2[0], fc2[1], None, x1, True)
    return x2.view(B, N, D)


class PatchEmbedTrainFn(torch.autograd.Function):
    """PatchEmbedFn with gradients for the projection weight/bias, the class token and the position embedding."""

    @staticmethod
    def forward(ctx, images, conv_w, conv_b, cls, pos, patch):
        dt = compute_dtype()
        B, C, H, W = images.shape
        G = (H // patch) * (W // patch)
        D = conv_w.shape[0]
        K = C * patch * patch
        gran = 64                                               # uia_wgrad needs both extents in multiples of 64
        Kp = (K + gran - 1) // gran * gran
        cols = _empty((B * G, Kp), dt, images)
        ops.im2col(images.contiguous().float(), cols, patch)
        x = torch.empty(B, G + 1, D, device=images.device, dtype=torch.float32)
        pos2d = pos.detach().reshape(G + 1, D).contiguous()
        ops.gemm(cols, WEIGHTS.get(conv_w, dt, pad_cols_to=Kp), bias=conv_b, resid=pos2d, resid_mod=G, resid_row_off=1, out_group=G, out32=x.view(-1, D))
        ops.fill_cls(x, cls.detach().reshape(D).contiguous(), pos2d[0])
        ctx.save_for_backward(cols)
        ctx.meta = (conv_w.shape, conv_b is not None, cls.shape, pos.shape, K, Kp)
        return x

    @staticmethod
    def backward(ctx, dx):
        (cols,) = ctx.saved_tensors
        wshape, has_bias, cshape, pshape, K, Kp = ctx.meta
        B, N, D = dx.shape
        G = N - 1
        dt = cols.dtype
        dx = dx.contiguous()
        dtok = dx[:, 1:, :].contiguous().view(B * G, D)
        dtok_t = dtok if dt == torch.float32 else torch.empty(B * G, D, device=dx.device, dtype=dt)
        if dt != torch.float32:
            ops.cast(dtok, dtok_t)
        dW = torch.zeros(D, Kp, device=dx.device, dtype=torch.float32)
        db = torch.zeros(D, device=dx.device, dtype=torch.float32) if has_bias else None
        ops.wgrad(dtok_t, cols, dW, db)
        dpos = torch.zeros(N * D, device=dx.device, dtype=torch.float32)
        ops.colsum(dx.view(B, N * D), dpos)                      # Σ over the batch: position-embedding gradient; row 0 is also d cls
        return None, dW[:, :K].contiguous().view(wshape), db, dpos[:D].clone().view(cshape), dpos.view(pshape), None


def trainable_cls_head(tokens, ln_w, ln_b, eps, proj_w):
    """feat = LN(tokens[:, 0]) @ proj_wᵀ with trainable norm and projection (timm head)."""
    cls = tokens[:, 0, :].contiguous()
    h = LayerNormAffineFn.apply(cls, ln_w, ln_b, eps)
    return LinearTrainFn.apply(CastFn.apply(h), proj_w, None, None, None, True)
