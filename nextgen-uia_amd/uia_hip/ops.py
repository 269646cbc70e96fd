"""Tensor-level wrappers over the C ABI: torch tensors in, kernels launched on the current stream.

PyTorch is only plumbing here (device memory, streams); there is no arithmetic in this file and no
CPU path: CPU tensors raise.  `dtype` of an op is taken from its operand tensors
(torch.bfloat16 -> UIA_BF16, torch.float32 -> UIA_F32).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import AttnDesc, GemmDesc, LnLoraDesc, LoraRankDesc, MonaFusedDesc, MonaSpatialDesc, PackDesc, UiaError, WgradGroupDesc, check, lib

_ACT = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "relu": 3}

# When set to a list, every uia_gemm launch is bracketed by two events on the launch stream and
# (start, end, M, N, K, dtype, tile_cfg) is appended: bench.py's live per-kernel roofline measurement.
GEMM_PROFILE = None


EPI_BIAS, EPI_AUX_OUT, EPI_GELU, EPI_DGELU, EPI_RESID, EPI_RESIDT, EPI_OUT32, EPI_OUTT, EPI_RESID_LN, EPI_GENERIC = 1, 2, 4, 8, 16, 32, 64, 128, 256, -1
EPI_QUICK = 2048
EPI_ROWSUM, EPI_LNFOLD = 512, 1024
EPI_RESID_LO, EPI_OUT_LO = 4096, 8192      # three-byte tensors (bf16 hi plane + signed low byte): uia_gemm_desc.resid_lo8 / out_lo8
_SPECIALISED = {EPI_BIAS | EPI_RESID | EPI_OUT32, EPI_BIAS | EPI_RESIDT | EPI_OUT32, EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUT32, EPI_OUTT, EPI_BIAS | EPI_OUTT, EPI_BIAS | EPI_GELU | EPI_OUTT, EPI_DGELU | EPI_OUTT,
                EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT,
                EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_OUTT, EPI_QUICK | EPI_DGELU | EPI_OUTT, EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT,
                EPI_BIAS | EPI_RESID | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM, EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM,
                EPI_BIAS | EPI_OUTT | EPI_LNFOLD, EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD, EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD,
                EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD, EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD,
                EPI_BIAS | EPI_RESID_LO | EPI_RESID_LN | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM,
                EPI_BIAS | EPI_RESID | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM, EPI_BIAS | EPI_RESID_LO | EPI_OUT32,
                EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM, EPI_BIAS | EPI_RESID_LO | EPI_RESID_LN | EPI_OUT32,
                EPI_BIAS | EPI_RESID_LO | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM}

# the masks csrc/gemm_quad.hip instantiates (tile cfg 25); the rest run its run-time epilogue
_QUAD_SPECIALISED = _SPECIALISED - {EPI_BIAS | EPI_RESIDT | EPI_OUT32, EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUT32, EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM,
                                    EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM, EPI_BIAS | EPI_RESID_LO | EPI_RESID_LN | EPI_OUT32,
                                    EPI_BIAS | EPI_RESID_LO | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM}


def epi_mask_of(d):
    """Mirror of epi_mask_of() in csrc/gemm.hip: the compile-time epilogue instantiation a descriptor lands on."""
    if d.alpha != 1.0 or d.out_group > 0 or d.resid_mod > 0:
        return EPI_GENERIC
    gl = (_ACT["gelu"], _ACT["quick_gelu"])
    if (d.act and d.act not in gl) or (d.dact and d.dact not in gl) or (d.act and d.dact and d.act != d.dact):
        return EPI_GENERIC
    quick = _ACT["quick_gelu"] in (d.act, d.dact)
    m = ((EPI_QUICK if quick else 0) | (EPI_BIAS if d.bias else 0) | (EPI_AUX_OUT if d.aux_out else 0) | (EPI_GELU if d.act else 0) | (EPI_DGELU if d.dact else 0) |
         (EPI_RESID if d.resid else 0) | (EPI_RESIDT if (d.residT and not d.resid_lo8) else 0) | (EPI_OUT32 if d.out32 else 0) | (EPI_OUTT if d.outT else 0) |
         (EPI_RESID_LN if ((d.resid or d.resid_lo8) and d.resid_ln_stats) else 0) | (EPI_ROWSUM if d.rowsum_out else 0) | (EPI_LNFOLD if d.lnfold_sums else 0) |
         (EPI_RESID_LO if d.resid_lo8 else 0) | (EPI_OUT_LO if d.out_lo8 else 0))
    return m


def auto_tile_cfg(M, N, K=None, esz=2, mask=EPI_GENERIC):
    """Mirror of launch_typed() in csrc/gemm.hip (which kernel instantiation a shape runs on)."""
    if N <= 64:
        stream64 = N == 64 and M > 2048 and esz == 2 and K is not None and K % 32 == 0 and 64 * (2 * K + 16) <= 160 * 1024
        if stream64 and mask in (EPI_OUTT, EPI_BIAS | EPI_OUTT, EPI_GENERIC):      # EPI_GENERIC = not known yet (the first look, before the descriptor is filled)
            return 16
        return 14 if (M > 2048 and esz == 2 and K is not None and K >= 1024) else 4
    if M <= 2048:
        return 21 if (esz == 4 and ((M + 127) // 128) * ((N + 127) // 128) < 64) else 3
    if K is not None and K * esz <= 128:
        return 14
    return 8


RING_CFGS = (8, 9, 10, 12, 13, 14, 24, 25, 26, 27, 28, 29)
PERSIST_STORE_ONLY = False   # experiment knob: 256x256 ring launches whose epilogue only stores T results run on the persistent variant (cfg 12)
KBLOCK_W = True          # hand the ring kernels their weights K-blocked (PackedW.kblocked()); False = row-major everywhere
K64_CFG14 = True         # single-K-step GEMMs on the two-workgroups-per-CU half-height config (False: 256x256 like every other large shape)
TILE_GROUP = {}          # experiment knob: {N: row panels per tile-order group} overriding the launcher's choice for ring launches with that N
SHORT_K_WIDE_HALF_STASH = True   # ... also for launches that write a second output (fc1 with the pre-activation stash)
SHORT_K_WIDE_HALF_BYTES = 2048   # ... and the longest K row (bytes) it applies to (fc1 of ViT-B: 1536, of ViT-L/14: 2048)
SHORT_K_WIDE_HALF_N = 3072   # N at and above which a short-K (K row <= SHORT_K_WIDE_HALF_BYTES) bf16 launch runs wholly on half-height tiles, two workgroups per CU (cfg 14); 0 = off (bench.py --short-k-half-n)
HALF_HEIGHT_SHORT_K_ALWAYS = False   # experiment knob: ... also when the 256-row tiling has no ragged last round (the text tower's N = 768, K = 768 sums at 65 536 rows)
HALF_HEIGHT_SHORT_K = True   # N <= 768, K <= 768 (bf16) launches whose 256-row tiling leaves a ragged last round run on half-height tiles (cfg 14) in one launch
CHAINS = 1               # independent chains of launches the caller has in flight on different streams (engine.contrastive_step sets it for the step): > 1 relaxes TAIL_SPLIT
TAIL_SPLIT = True        # split off the M tail of a launch whose last round of 256x256 tiles would leave most CUs idle
TAIL_SIDE_STREAM = True  # a SMALL M tail (at most a quarter of the CUs' worth of half-height tiles) runs on a side stream beside the main launch instead of behind it
MONA_PRE_FWD_T = True    # Mona forward: project1 (768 -> 64) inside the pre-norm row kernel (uia_mona_pre_fwd_t) instead of the N = 64 stream launch reading u back
MONA_PRE_BWD_DU = True   # Mona backward: project1's data gradient (K = 64) inside the pre-norm backward row kernel (uia_mona_pre_bwd_du) instead of a GEMM launch + a 77 MB round trip
LN_LORA_DOWN = True      # LoRA block: LayerNorm and the q / k / v down-projections drop_i(h)·A_iᵀ in one launch (uia_ln_lora_down) instead of the LayerNorm kernel + three N = 64 launches reading h back
LORA_REGEN_DROP = True   # LoRA input dropout: the forward does not write the dropped rows; the dA weight-gradient launch regenerates the mask while it stages x (uia_wgrad_drop)
LORA_RANK3 = True        # q | k | v of a LoRA block: the three rank terms of the data gradient in one pass over it (uia_lora_rank_update) instead of three K = 64 launches
QUAD = False             # 256x256 bf16 launches on the four-wave kernel (tile cfg 25, csrc/gemm_quad.hip: 128 x 128 per wave, one wave per SIMD) instead of cfg 8
QUADV = False            # experiment knob: 256x256 bf16 launches on the four-wave register-staged kernel (tile cfg 27, csrc/gemm_quadv.hip) instead of cfg 8
RING5 = False            # experiment knob: 256x256 launches with a long K loop or a wide N on the 5-deep ring (tile cfg 24: 160 KB of LDS, four sub-tiles in flight)
LORA_KEXT = True         # LoraAttnHalfFn: the rank update inside the frozen GEMM's K loop (uia_gemm_desc.A2 / K2) instead of a read-modify-write launch of its own
TAIL_SPLIT_K = True      # ... and run that tail split over K when it is a few tiles with a long K chain (two launches: slice partials, then sum + epilogue).
                         # Its slices meet through hardware float atomics: the rows of such a tail (<= 1/4 of the CUs busy: the ViT-L/14 + LoRA step, not the
                         # headline) vary in the last bit from run to run; set_deterministic(True) in functional turns it off.
_SPLITK_WS = {}
_TRACE_GENERIC = {} if __import__("os").environ.get("UIA_TRACE_GENERIC") else None
_TAIL_SIDE = {}
_NCU = {}


def num_cus(device=None):
    dev = torch.cuda.current_device() if device is None else device
    if dev not in _NCU:
        _NCU[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    return _NCU[dev]


def big_tile_cfg(N, K, esz):
    """8 or 24 for a launch auto_tile_cfg() puts on the 256x256 ring tiles.  Isolated, plain epilogue (tools/time_ring_depth.py) the fifth ring slot is
    worth 4-5.5 % at K = 3072 and at 43 520 rows and 0.5-1.6 % at 65 536 x {2304, 3072} x 768 (and costs 7 % at N = 768, K = 768); inside the step, behind the
    fused epilogues, it is level to slightly worse on every one of those launches (44.5-44.9 ms either way, three alternating pairs on one box): opt-in."""
    if QUAD and esz == 2:
        return 25
    if QUADV and esz == 2:
        return QUADV if QUADV in (27, 29) else 27
    return 24 if (RING5 and esz == 2 and (K * esz >= 2048 or N >= 2304)) else 8


def tail_k_slices(rows, N, K, esz, ncu):
    """K slices for the half-height tail launch (tile cfg 13) of `rows` rows, or 0.  A tail of t tiles keeps t of the CUs busy for the whole K chain
    (ViT-L/14's 128-row tails: 4 tiles, 64 us at K = 4096); sliced over K it uses s·t CUs for 1/s of the chain, at the price of a second launch
    that sums the slices.  Worth it for at most a quarter of the CUs' worth of tiles and from 4 KiB of K per row; at least 1 KiB of K per slice."""
    tiles = -(-rows // 128) * -(-N // 256)
    kb = K * esz
    if not TAIL_SPLIT_K or esz != 2 or kb < 4096 or 4 * tiles > ncu:           # 81 tiles x 3 slices (the headline step's tails) measured level: the sum launch eats the gain
        return 0
    s = min(ncu // tiles, kb // 1024, 16)
    return s if s >= 2 else 0


def splitk_workspace(floats, device):
    """fp32 scratch of the split-K tail launches: one per (device, stream) — consecutive launches of a stream are ordered, two streams are not."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    t = _SPLITK_WS.get(key)
    if t is None or t.numel() < floats:
        t = _SPLITK_WS[key] = torch.zeros(max(floats, 1 << 20), device=device, dtype=torch.float32)     # zero once: the launches leave it zero
    return t


class _TailFork:
    """Fork / join around the tail launches of a split GEMM: `with fork:` enqueues on a side stream that starts where the caller's stream stands;
    fork.join() makes the caller's stream wait for it.  ViT-L/14 at 128 pairs has M = 128·257 rows: 128 full row panels (exactly two rounds of
    256 x 256 tiles at N = 1024) and ONE half panel, whose 4-16 tiles ran for 16 us per launch (x 2 with split K) BEHIND a 75-300 us main launch —
    pure latency, 2.3 ms per step.  Beside the main launch they cost a few CUs a few microseconds."""

    def __init__(self, device):
        self.cur = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device(), self.cur.cuda_stream)
        e = _TAIL_SIDE.get(key)
        if e is None:
            e = _TAIL_SIDE[key] = (torch.cuda.Stream(device=device), torch.cuda.Event(), torch.cuda.Event())
        self.side, self.ev_in, self.ev_out = e
        self.ctx = None

    def __enter__(self):
        self.ev_in.record(self.cur)
        self.side.wait_event(self.ev_in)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ev_out.record(self.side)
        self.ctx.__exit__(*exc)
        return False

    def join(self):
        self.cur.wait_event(self.ev_out)


def small_tail(rows, N, ncu):
    """True when the tail of `rows` rows is few enough half-height tiles to run beside the main launch (TAIL_SIDE_STREAM)."""
    return TAIL_SIDE_STREAM and 4 * (-(-rows // 128) * -(-N // 256)) <= ncu


def drop_splitk_workspace(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    _SPLITK_WS.pop(key, None)


def tail_split_rows(M, N, ncu, bm=256, bn=256):
    """Rows of the MAIN launch (a multiple of `bm`) when the 256x256 tiling should be split, else M.

    tiles = ceil(M/bm)·ceil(N/bn) workgroups run in rounds of `ncu`; when the last round holds at most half of the CUs
    (M = 50 432, N = 768: 591 = 2·256 + 79) its tiles are re-cut at half height (tile config 13, 128x256) so that they spread over
    twice as many CUs and the round takes about half as long: 2.55 instead of 3 tile times for that shape.  Pure scheduling —
    every output element is still one K-ordered fp32 accumulation, so results are identical to the unsplit launch."""
    tm, tn = -(-M // bm), -(-N // bn)
    tiles = tm * tn
    rem = tiles % ncu
    if tiles <= ncu or rem == 0 or 2 * rem > ncu:
        return M
    main_panels = (tiles - rem) // tn
    if main_panels < 1 or main_panels >= tm:
        return M
    return main_panels * bm


ROWSUM_SCALE = float(1 << 30)    # row sums (uia_gemm rowsum / lnfold / resid_ln with dim) are int64 fixed point in units of 2^-30: exact, order-free atomics


def rowsum_from_float(s):
    """fp32 [M, 2] (Σ, Σ²) -> the fixed-point form the kernels exchange (tests and host-side producers)"""
    return (s.double() * ROWSUM_SCALE).round().to(torch.int64).contiguous()


def rowsum_to_float(q):
    return (q.double() / ROWSUM_SCALE).float()


# Guard word of the folded LayerNorms (uia_gemm_desc.ln_flag): one zeroed int32 per device, handed to EVERY launch that writes or reads
# row sums, so every folded layer of every step is checked (on the device, at the cost of a compare per row) — not just the first few
# buffers of a warm-up step.  uia_hip.functional.poll_ln_flag() reads it back.
LN_FLAG_LIMIT = 8.0
_LN_FLAG = {}


def ln_flag(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _LN_FLAG.get(key)
    if t is None:
        t = _LN_FLAG[key] = torch.zeros(1, device=device, dtype=torch.int32)
    return t


KBLOCK_ACT = True        # GEMM -> GEMM activations travel K-blocked ([K/g][rows][g]) between ring-kernel launches (kb_empty / is_kb)


def kb_group(dt):
    return 64 // torch.empty(0, dtype=dt).element_size()


def kb_ok(M, N_consumer, K, dt):
    """May an [M, K] activation that ONE GEMM epilogue writes and ONE GEMM (N_consumer columns) reads travel K-blocked?  Both launches
    must land on a ring tile config with 64-byte sub-tiles (M > 2048 and more than 64 columns on either side)."""
    return (KBLOCK_ACT and K % kb_group(dt) == 0 and K > 64 and auto_tile_cfg(M, N_consumer, K, torch.empty(0, dtype=dt).element_size()) in (8, 14)
            and auto_tile_cfg(M, K, 128, torch.empty(0, dtype=dt).element_size()) == 8)


class KBlocked:
    """An [M, K] activation stored K-blocked: `.t` is a [K/g, M, g] tensor (g = 64 bytes of elements; a row range `.t[:, lo:hi]` of a
    larger one is fine).  An explicit type rather than "any 3-D tensor": only ops.gemm (a / out_t) and the few producer kernels that
    were taught the layout accept it, everything else fails loudly on it."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t

    dtype = property(lambda self: self.t.dtype)
    device = property(lambda self: self.t.device)
    is_cuda = property(lambda self: self.t.is_cuda)
    rows = property(lambda self: self.t.shape[1])
    cols = property(lambda self: self.t.shape[0] * self.t.shape[2])

    def element_size(self):
        return self.t.element_size()

    def data_ptr(self):
        return self.t.data_ptr()

    def row_range(self, lo, hi):
        return KBlocked(self.t[:, lo:hi])

    def as_rows(self):
        """the same STORAGE as a row-major [M, K] buffer (contents are not converted: for re-use of a dead buffer)"""
        return self.t.reshape(-1).view(self.rows, self.cols)

    @staticmethod
    def over(buf):
        """the storage of a dead row-major [M, K] buffer as a K-blocked one (contents are not converted)"""
        g = kb_group(buf.dtype)
        M, K = buf.shape
        return KBlocked(buf.view(K // g, M, g))


def kb_empty(M, K, dt, device):
    g = kb_group(dt)
    return KBlocked(torch.empty(K // g, M, g, device=device, dtype=dt))


def is_kb(t):
    return isinstance(t, KBlocked)


def _kb_dims(kb, name):
    t = kb.t
    g = kb_group(t.dtype)
    if t.shape[2] != g or t.stride(2) != 1 or t.stride(1) != g or t.stride(0) % g or t.stride(0) // g < t.shape[1]:
        raise UiaError(f"{name}: not a K-blocked [K/{g}, rows, {g}] activation (shape {tuple(t.shape)}, stride {t.stride()})")
    return t.shape[1], t.shape[0] * g, t.stride(0) // g          # rows, columns, rows of the whole tensor (plane stride)


class Ready:
    """The stream that filled a cached device buffer and the event that orders every OTHER stream behind the fill (the weight
    caches are filled by whichever stream misses first; a hit from another stream — contrastive_step(streams=S), the text tower's side
    stream — launched GEMMs on a buffer whose cast / pack / transpose could still be in flight).  sync() is a set lookup on the filling
    stream and on every stream that has waited once."""
    __slots__ = ("event", "seen")

    def __init__(self):
        st = torch.cuda.current_stream()
        self.event = torch.cuda.Event()
        self.event.record(st)
        self.seen = {st.cuda_stream}

    def sync(self):
        rs = raw_stream()
        if rs not in self.seen:
            torch.cuda.current_stream().wait_event(self.event)
            self.seen.add(rs)


class PackedW:
    """A GEMM weight [N, K] in the compute dtype, with its K-blocked twin [K/g][N][g] (g = 64 bytes of elements) built on first
    use by a ring-kernel launch.  Layout plumbing only (a strided copy); which one a launch takes is decided in gemm().
    Construct it AFTER the kernels that fill `row` (and `kb`) were enqueued: it records the event other streams wait for."""
    __slots__ = ("_row", "_kb", "ready", "kb_ready")

    def __init__(self, row, kb=None):
        self._row, self._kb = row, kb
        self.ready = Ready() if row.is_cuda else None
        self.kb_ready = self.ready

    def refreshed(self, ready):
        """the buffers were rewritten in place (uia_pack_weights after an optimiser step)"""
        self.ready = self.kb_ready = ready

    @property
    def row(self):
        if self.ready is not None:
            self.ready.sync()
        return self._row

    @property
    def shape(self):
        return self._row.shape

    @property
    def dtype(self):
        return self._row.dtype

    @property
    def device(self):
        return self._row.device

    def kblocked(self):
        if self._kb is None:
            row = self.row                                       # orders this stream behind the fill of `row`
            N, K = row.shape
            g = 64 // row.element_size()
            self._kb = row.view(N, K // g, g).permute(1, 0, 2).contiguous()
            self.kb_ready = Ready() if row.is_cuda else None
        elif self.kb_ready is not None:
            self.kb_ready.sync()
        return self._kb


class ExtW:
    """A GEMM weight [N, K] that exists K-BLOCKED only ([K/g][N][g]) and whose last K2 columns pair with a second A operand (uia_gemm_desc.A2 / K2: the LoRA
    rank update inside the frozen GEMM).  Built by functional.WEIGHTS.get_lora_ext."""
    __slots__ = ("_kb", "N", "K", "K2", "ready", "__weakref__")

    def __init__(self, kb, N, K, K2):
        self._kb, self.N, self.K, self.K2 = kb, N, K, K2
        self.ready = Ready() if kb.is_cuda else None

    @property
    def kb(self):
        if self.ready is not None:
            self.ready.sync()
        return self._kb

    @property
    def device(self):
        return self._kb.device


class RawDest:
    """A uia_pack_weights destination that is a strided view inside a larger buffer (the kernel's rows_pad / cols_pad reproduce its strides)."""
    __slots__ = ("view",)

    def __init__(self, view):
        self.view = view


def gemm_kernel_name(cfg, mask, dtype):
    """(readable name, mangled fragment) of the instantiation a launch runs on.  rocprofv3 prints these kernels mangled
    (its demangler does not know the bf16 type code), so the fragment is what to grep for in profiles/*.csv."""
    m = mask if mask in _SPECIALISED else EPI_GENERIC
    tn, tc = ("bf16", "DF16b") if dtype == torch.bfloat16 else ("float", "f")
    mi = lambda v: f"Li{v}E" if v >= 0 else f"Lin{-v}E"
    if cfg == 12:
        return f"gemm_tn_persist_kernel<{tn},{m}>", f"gemm_tn_persist_kernelI{tc}{mi(m)}E"
    if cfg in (8, 13, 14, 24):
        bm, nbuf = (256, 4) if cfg == 8 else ((256, 5) if cfg == 24 else ((128, 4) if cfg == 13 else (128, 3)))
        return f"gemm_tn_ring_kernel<{tn},{bm},256,2,4,64,{nbuf},{m}>", "gemm_tn_ring_kernelI" + tc + "".join(mi(v) for v in (bm, 256, 2, 4, 64, nbuf, m, 0)) + "Lb0ELb0EE"
    if cfg in (25, 26):
        mq = m if (cfg == 25 and m in _QUAD_SPECIALISED) else EPI_GENERIC
        return f"gemm_tn_quad_kernel<{mq},false>", f"gemm_tn_quad_kernelI{mi(mq)}Lb0EE"
    if cfg == 29:
        mq = m if m in _QUAD_SPECIALISED else EPI_GENERIC
        return f"gemm_tn_quadvp_kernel<{mq}>", f"gemm_tn_quadvp_kernelI{mi(mq)}E"
    if cfg in (27, 28):
        mq = m if (m in _QUAD_SPECIALISED and (cfg == 27 or m == EPI_OUTT)) else EPI_GENERIC
        return f"gemm_tn_quadv_kernel<{mq},{cfg - 25},0>", f"gemm_tn_quadv_kernelI{mi(mq)}{mi(cfg - 25)}Li0EE"
    if cfg == 16:
        return "gemm_skinny64_kernel", "gemm_skinny64_kernel"
    if cfg == 23:
        return "gemm_wide64_kernel", "gemm_wide64_kernel"
    shape = {1: (256, 256, 2, 4), 2: (256, 128, 4, 2), 3: (128, 128, 2, 2), 4: (256, 64, 4, 1), 5: (128, 64, 2, 1), 21: (32, 64, 2, 2)}.get(cfg)
    if shape:
        return f"gemm_tn_kernel<{tn},{','.join(map(str, shape))}>", "gemm_tn_kernelI" + tc + "".join(mi(v) for v in shape) + "E"
    return f"tile cfg {cfg}", f"cfg{cfg}"


def _code(dt):
    if dt == torch.bfloat16:
        return _lib.BF16
    if dt == torch.float32:
        return _lib.F32
    raise UiaError(f"unsupported operand dtype {dt}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise UiaError("uia ops need CUDA/HIP tensors: there is no CPU fallback for the hot path")
    return t.data_ptr()


_HIP_OK = []


def raw_stream():
    """Handle of the current stream of the current device.  torch.cuda.current_stream() costs ~8 us of Python per call (device-index helpers, is_available, an
    os.environ lookup, a Stream object); every launch and every cached-weight hit asks — 390 times per CLIPSeg step, 3 ms of an 11 ms host-bound step.  Two C calls."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    if not _HIP_OK:
        if not torch.cuda.is_available():
            raise UiaError("uia ops need an MI355X (HIP) device: there is no CPU fallback for the hot path")
        _HIP_OK.append(True)
    return raw_stream()


def _rowmajor(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise UiaError(f"{name} must be a 2-D row-major tensor (got shape {tuple(t.shape)}, stride {t.stride()})")
    return t.stride(0)


def gemm(a, w, *, bias=None, act=None, dact=None, aux_in=None, aux_out=None, resid=None, resid_mod=0, resid_row_off=0,
         resid_t=None, out_group=0, out_t=None, out32=None, alpha=1.0, tile_cfg=0, resid_ln=None, rowsum=None, lnfold=None, drop=None, a2=None,
         resid3=None, out_lo=None):
    """C = epilogue(alpha * a @ w.T).  a [M,K], w [N,K] (tensor or PackedW) share a dtype (bf16 | fp32); see include/uia_hip.h.

    Host-side scheduling on top of uia_gemm (results do not depend on it): with the automatic tile choice, a weight that came
    as a PackedW is handed to the ring kernels K-blocked, and the M tail of a launch whose last round of tiles would leave most
    CUs idle goes through a second launch with half-height tiles (tail_split_rows).
    resid_ln = (stats [M, 2], ln_weight, ln_bias): `resid` is the INPUT of a LayerNorm whose output is the residual to add; a fourth and
    fifth element (dim, eps) say that `stats` holds the row sums (Σ, Σ²) a producing GEMM left through `rowsum`, not (mean, rstd).
    drop = ("a", p, seed[, a_drop_out]): LoRA input dropout applied to `a` in flight (N = 64 stream kernel only; a_drop_out receives the dropped
    rows), or ("acc", p, seed): applied to alpha·acc of element (m, n) before the residual adds — the generator of ops.dropout in both cases.
    rowsum = zeroed int64 [M, 2] (ROWSUM_SCALE fixed point; rowsum_to_float converts): receives (Σ, Σ²) of the stored fp32 rows.  lnfold = (sums [M, 2], colsum [N], dim, eps): `a` holds RAW
    rows and `w` is pre-scaled by the LayerNorm weight; the epilogue applies the LayerNorm (include/uia_hip.h, uia_gemm_desc).
    Three-byte tensors (bf16, ring tile configs: M > 2048): resid3 = (hi, lo) is the residual as a bf16 hi plane (row-major or KBlocked) plus an int8
    [M, N] plane of low bytes — float bits = (hi_bits << 16) + (lo << 8) — to which resid_ln then applies; out_lo = int8 [M, N] receives the low
    bytes of the result, whose hi plane is out_t (three_byte_to_float converts)."""
    if (resid3 is not None or out_lo is not None) and (isinstance(w, ExtW) or resid_mod or out_group or (a.t.dtype if is_kb(a) else a.dtype) != torch.bfloat16):
        raise UiaError("gemm: three-byte tensors (resid3 / out_lo) need bf16 operands and a plain epilogue (no row remapping, no K extension)")
    if isinstance(w, ExtW):
        # K extension: w = [W | s·B] (K-blocked only), a2 = (t, group_cols): t [M, K2] or [G, M, K2] — one [M, K2] operand per group of output columns
        if a2 is None or is_kb(a) or a.dtype != torch.bfloat16 or a.shape[1] != w.K - w.K2 or resid_mod or out_group or drop is not None:
            raise UiaError("gemm: an ExtW weight needs a2 = (t, group_cols), a row-major bf16 `a` of K - K2 columns and a plain epilogue")
        t2, gcols = a2
        M = a.shape[0]
        if t2.dtype != a.dtype or t2.shape[-1] != w.K2 or t2.shape[-2] != M or t2.stride(-1) != 1 or t2.stride(-2) != w.K2 or (t2.dim() == 3 and t2.shape[0] * gcols != w.N):
            raise UiaError(f"gemm: a2 operand {tuple(t2.shape)} does not match [groups, {M}, {w.K2}] with {w.N} columns in groups of {gcols}")
        m_main = tail_split_rows(M, w.N, num_cus(a.device.index)) if (TAIL_SPLIT and tile_cfg == 0 and M > 2048) else M
        cut = lambda t, lo, hi: None if t is None else t[lo:hi]
        def part(lo, hi, cfg):
            _gemm_one(a[lo:hi], w, bias=bias, act=act, dact=dact, aux_in=cut(aux_in, lo, hi), aux_out=cut(aux_out, lo, hi), resid=cut(resid, lo, hi),
                      resid_t=cut(resid_t, lo, hi), out_t=cut(out_t, lo, hi), out32=cut(out32, lo, hi), alpha=alpha, tile_cfg=cfg,
                      resid_ln=None if resid_ln is None else (resid_ln[0][lo:hi],) + tuple(resid_ln[1:]), rowsum=cut(rowsum, lo, hi),
                      lnfold=None if lnfold is None else (lnfold[0][lo:hi],) + tuple(lnfold[1:]), a2=(t2[..., lo:hi, :], gcols))
        if m_main >= M:
            part(0, M, tile_cfg or (8 if M > 2048 else 13))
        elif a.is_cuda and small_tail(M - m_main, w.N, num_cus(a.device.index)):
            fork = _TailFork(a.device)
            with fork:
                part(m_main, M, 13)
            part(0, m_main, 8)
            fork.join()
        else:
            part(0, m_main, 8)
            part(m_main, M, 13)
        return
    packed = w if isinstance(w, PackedW) else None
    wrow = packed.row if packed is not None else w
    M, N = (a.rows if is_kb(a) else a.shape[0]), wrow.shape[0]
    Ka = wrow.shape[1]
    if not K64_CFG14 and tile_cfg == 0 and auto_tile_cfg(M, N, Ka, a.element_size()) == 14:
        tile_cfg = 8
    if (TAIL_SPLIT and tile_cfg == 0 and out_group == 0 and resid_mod == 0 and a.is_cuda and drop is None and auto_tile_cfg(M, N, Ka, a.element_size()) == 8):
        m_main = tail_split_rows(M, N, num_cus(a.device.index))
        if (SHORT_K_WIDE_HALF_N and N >= SHORT_K_WIDE_HALF_N and Ka * a.element_size() <= SHORT_K_WIDE_HALF_BYTES and a.dtype == torch.bfloat16
                and (SHORT_K_WIDE_HALF_STASH or aux_out is None)):
            # short K loop, wide N (fc1 and its GELU' data gradient: K = 768, N = 3072): a quarter of a 256 x 256 launch is prologue + epilogue + hand-over that
            # nothing overlaps (profiles/r04_d); on half-height tiles two workgroups share a CU and one's seam runs beside the other's K loop:
            # 385 -> 338 us at M = 65 536, 266 -> 252 at 50 432 in isolation (bias + GELU store), level at N = 2304
            tile_cfg, m_main = 14, M
        if (m_main < M or HALF_HEIGHT_SHORT_K_ALWAYS) and HALF_HEIGHT_SHORT_K and CHAINS <= 1 and N <= 768 and Ka * a.element_size() <= 1536:      # (with other chains beside it a ragged round costs nothing: 40.99 -> 40.73 ms without this rule there)
            # short K loops with a ragged last round (the image tower's output projection and its data gradient: 591 tiles on 256 CUs): the whole
            # launch on half-height tiles, two workgroups per CU (tile cfg 14), instead of a main launch + a half-height tail launch — 1182 half
            # tiles pack 2.31 rounds of 512, and one workgroup's epilogue runs beside its neighbour's K loop (round 3, isolated: 64 vs 62 + 21 us
            # for the plain data gradient, 92 vs 102 us per 43 520 rows for the fp32-residual producer).  Longer K loops lose on it.
            tile_cfg, m_main = 14, M
        if m_main < M and CHAINS > 1 and not small_tail(M - m_main, N, num_cus(a.device.index)):
            # several independent chains of launches share the chip (engine.contrastive_step: the image tower in two slices beside the text tower): a ragged last
            # round is filled by the other chains' workgroups, and a tail launch BEHIND the main one only adds a launch (40.87 -> 40.70 ms).  Small tails still
            # fork to the side stream: without that split the ViT-L/14 + LoRA step loses 2.4 ms (a handful of tiles as a whole extra round).
            m_main = M
        if m_main < M:
            cut = lambda t, lo, hi: None if t is None else (t.row_range(lo, hi) if is_kb(t) else t[lo:hi])
            slices = tail_k_slices(M - m_main, N, Ka, a.element_size(), num_cus(a.device.index))

            def part(lo, hi, cfg, ws=None):
                _gemm_one(cut(a, lo, hi), w, bias=bias, act=act, dact=dact, aux_in=cut(aux_in, lo, hi), aux_out=cut(aux_out, lo, hi), resid=cut(resid, lo, hi),
                          resid_t=cut(resid_t, lo, hi), out_t=cut(out_t, lo, hi), out32=cut(out32, lo, hi), alpha=alpha, tile_cfg=cfg,
                          resid_ln=None if resid_ln is None else (resid_ln[0][lo:hi],) + tuple(resid_ln[1:]), rowsum=cut(rowsum, lo, hi),
                          lnfold=None if lnfold is None else (lnfold[0][lo:hi],) + tuple(lnfold[1:]), splitk_ws=ws if cfg >> 16 else None,
                          resid3=None if resid3 is None else (cut(resid3[0], lo, hi), cut(resid3[1], lo, hi)), out_lo=cut(out_lo, lo, hi))

            def tail():      # on whichever stream is current: the split-K scratch belongs to (device, stream)
                ws = splitk_workspace(-(-(M - m_main) // 128) * -(-N // 256) * 128 * 256, a.device) if slices else None
                try:
                    for cfg in ((13 | slices << 16 | 1 << 22, 13 | slices << 16 | 2 << 22) if slices else (13,)):
                        part(m_main, M, cfg, ws)
                except Exception:
                    if slices:      # phase 2 is what re-zeroes the shared scratch: an error between the two launches must not leave partial sums behind
                        drop_splitk_workspace(a.device)
                    raise

            if small_tail(M - m_main, N, num_cus(a.device.index)):
                fork = _TailFork(a.device)
                with fork:
                    tail()
                part(0, m_main, big_tile_cfg(N, Ka, a.element_size()))
                fork.join()
            else:
                part(0, m_main, big_tile_cfg(N, Ka, a.element_size()))
                tail()
            return
    _gemm_one(a, w, bias=bias, act=act, dact=dact, aux_in=aux_in, aux_out=aux_out, resid=resid, resid_mod=resid_mod, resid_row_off=resid_row_off,
              resid_t=resid_t, out_group=out_group, out_t=out_t, out32=out32, alpha=alpha, tile_cfg=tile_cfg, resid_ln=resid_ln,
              rowsum=rowsum, lnfold=lnfold, drop=drop, resid3=resid3, out_lo=out_lo)


def _gemm_one(a, w, *, bias=None, act=None, dact=None, aux_in=None, aux_out=None, resid=None, resid_mod=0, resid_row_off=0,
              resid_t=None, out_group=0, out_t=None, out32=None, alpha=1.0, tile_cfg=0, resid_ln=None, rowsum=None, lnfold=None, drop=None, splitk_ws=None, a2=None,
              resid3=None, out_lo=None):
    d = GemmDesc()
    if splitk_ws is not None:
        d.splitk_ws = _p(splitk_ws)
    ext = w if isinstance(w, ExtW) else None
    if ext is not None:
        t2, gcols = a2
        d.A2, d.lda2, d.K2 = t2.data_ptr(), t2.stride(-2), ext.K2
        d.a2_group_cols, d.a2_group_stride = (gcols, t2.stride(0)) if t2.dim() == 3 else (0, 0)
    if drop is not None:
        d.drop_where, d.drop_p, d.drop_seed = {"a": 1, "acc": 2}[drop[0]], float(drop[1]), int(drop[2]) & 0xFFFFFFFFFFFFFFFF
        if len(drop) > 3 and drop[3] is not None:
            xo = drop[3]
            if drop[0] != "a" or xo.dtype != a.dtype or xo.shape != a.shape or xo.stride() != a.stride():
                raise UiaError("gemm drop: a_drop_out must have a's dtype, shape and strides")
            d.a_drop_out = _p(xo)
    packed = w if isinstance(w, PackedW) else None
    if packed is not None:
        w = packed.row
    if ext is not None:
        # W exists K-blocked only and spans K = (columns of a) + K2; the launch runs on the ring tiles it was given (8 / 13)
        d.lda, d.ldw, d.M, d.K, d.N = _rowmajor(a, "a"), ext.K, a.shape[0], ext.K, ext.N
        d.A, d.W, d.w_kblocked = _p(a), _p(ext.kb), 1
        base_cfg = (tile_cfg & 255) or (8 if d.M > 2048 else 13)
        tile_cfg = (tile_cfg & ~255) | base_cfg
    else:
        d.ldw = _rowmajor(w, "w")
        if is_kb(a):
            d.M, d.K, d.a_kb_rows = _kb_dims(a, "gemm a")
            d.lda = d.K
        else:
            d.lda = _rowmajor(a, "a")
            d.M, d.K = a.shape[0], a.shape[1]
        if a.dtype != w.dtype or d.K != w.shape[1]:
            raise UiaError(f"gemm operand mismatch: a {(d.M, d.K)} {a.dtype}, w {tuple(w.shape)} {w.dtype}")
        d.A, d.W = _p(a.t if is_kb(a) else a), _p(w)
        d.N = w.shape[0]
        base_cfg = (tile_cfg & 255) or auto_tile_cfg(d.M, d.N, d.K, a.element_size())
    # mirror of wide64_ok() in csrc/gemm.hip: the K = 64 read-modify-write stream (LoRA rank update / its data gradient) takes row-major W
    if ((tile_cfg & 255) == 0 and base_cfg == 14 and a.dtype == torch.bfloat16 and d.K == 64 and d.N % 64 == 0 and d.N * 144 <= 160 * 1024 and not is_kb(a)
            and not is_kb(out_t) and act is None and dact is None and aux_out is None and out_group == 0 and resid_mod == 0 and rowsum is None and lnfold is None
            and resid_ln is None and (drop is None or drop[0] == "acc")
            and ((out_t is not None and out32 is None and resid is None and (resid_t is not None or drop is not None))
                 or (out32 is not None and out_t is None and resid_t is None and (resid is not None or drop is not None)))):
        base_cfg = 23
    if (tile_cfg & 255) == 0 and base_cfg == 8:
        base_cfg = big_tile_cfg(d.N, d.K, a.element_size())
        tile_cfg |= base_cfg
    if (PERSIST_STORE_ONLY and base_cfg == 8 and resid is None and resid_t is None and out32 is None and rowsum is None and out_group == 0
            and alpha == 1.0 and d.K * a.element_size() >= 1024):
        base_cfg, tile_cfg = 12, (tile_cfg & ~255) | 12
    if packed is not None and KBLOCK_W and base_cfg in RING_CFGS and a.is_cuda:
        d.W, d.w_kblocked = _p(packed.kblocked()), 1
    d.alpha = alpha
    d.act, d.dact = _ACT[act], _ACT[dact]
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == d.N
        d.bias = _p(bias)
    for name, t, want in (("aux_in", aux_in, a.dtype), ("aux_out", aux_out, a.dtype), ("resid_t", resid_t, a.dtype),
                          ("out_t", out_t, a.dtype), ("resid", resid, torch.float32), ("out32", out32, torch.float32)):
        if t is not None and t.dtype != want:
            raise UiaError(f"gemm {name} must be {want}, got {t.dtype}")
    if is_kb(out_t):
        rows, cols, plane = _kb_dims(out_t, "gemm out_t")
        if rows < d.M or cols != d.N or out_group:
            raise UiaError(f"gemm out_t (K-blocked {tuple(out_t.t.shape)}) does not hold the [{d.M}, {d.N}] result")
    for name, t in (("aux_in", aux_in), ("aux_out", aux_out), ("resid_t", resid_t), ("out_t", None if is_kb(out_t) else out_t), ("out32", out32)):
        if t is not None and (t.dim() != 2 or t.shape[1] < d.N or (t.shape[0] < d.M and out_group == 0)):
            raise UiaError(f"gemm {name} is {tuple(t.shape)}: too small for the [{d.M}, {d.N}] result")
    if resid is not None and resid_mod == 0 and out_group == 0 and (resid.dim() != 2 or resid.shape[1] < d.N or resid.shape[0] < d.M):
        raise UiaError(f"gemm resid is {tuple(resid.shape)}: too small for the [{d.M}, {d.N}] result")
    if aux_in is not None:
        d.aux_in, d.ldaux_in = _p(aux_in), _rowmajor(aux_in, "aux_in")
    if aux_out is not None:
        d.aux_out, d.ldaux_out = _p(aux_out), _rowmajor(aux_out, "aux_out")
    if resid is not None:
        d.resid, d.ldr = _p(resid), _rowmajor(resid, "resid")
    d.resid_mod, d.resid_row_off, d.out_group = resid_mod, resid_row_off, out_group
    if resid_ln is not None:
        st, lw, lb = resid_ln[:3]
        if len(resid_ln) > 3:
            d.resid_ln_dim, d.resid_ln_eps = int(resid_ln[3]), float(resid_ln[4])
        if (resid is None and resid3 is None) or resid_mod or out_group:
            raise UiaError("gemm resid_ln needs a plain fp32 resid or a three-byte resid3 (no row remapping)")
        want_st = torch.int64 if len(resid_ln) > 3 else torch.float32       # row sums (fixed point) or (mean, rstd)
        if not (st.dtype == want_st and lw.dtype == lb.dtype == torch.float32 and st.is_contiguous() and st.numel() >= 2 * d.M and lw.numel() >= d.N and lb.numel() >= d.N):
            raise UiaError(f"gemm resid_ln: stats {tuple(st.shape)} / weight {tuple(lw.shape)} / bias {tuple(lb.shape)} do not cover [{d.M}, {d.N}]")
        d.resid_ln_stats, d.resid_ln_w, d.resid_ln_b = _p(st), _p(lw), _p(lb)
    if rowsum is not None:
        if not (rowsum.dtype == torch.int64 and rowsum.is_contiguous() and rowsum.numel() >= 2 * d.M) or out_group:
            raise UiaError(f"gemm rowsum must be a contiguous int64 [{d.M}, 2] tensor (no row remapping), got {tuple(rowsum.shape)} {rowsum.dtype}")
        d.rowsum_out = _p(rowsum)
    if lnfold is not None:
        sm, cs, dim, eps = lnfold
        if not (sm.dtype == torch.int64 and cs.dtype == torch.float32 and sm.is_contiguous() and cs.is_contiguous() and sm.numel() >= 2 * d.M and cs.numel() >= d.N) or alpha != 1.0:
            raise UiaError(f"gemm lnfold: sums {tuple(sm.shape)} {sm.dtype} / colsum {tuple(cs.shape)} do not cover [{d.M}, {d.N}] (int64 row sums, fp32 colsum, contiguous, alpha == 1)")
        d.lnfold_sums, d.lnfold_colsum, d.lnfold_dim, d.lnfold_eps = _p(sm), _p(cs), int(dim), float(eps)
    if (rowsum is not None or lnfold is not None) and a.is_cuda:
        d.ln_flag, d.ln_flag_limit = _p(ln_flag(a.device)), LN_FLAG_LIMIT
    if resid_t is not None:
        d.residT, d.ldrT = _p(resid_t), _rowmajor(resid_t, "resid_t")
    if resid3 is not None:
        hi3, lo3 = resid3
        if resid is not None or resid_t is not None:
            raise UiaError("gemm: resid3 replaces resid / resid_t")
        if is_kb(hi3):
            rows, cols, plane = _kb_dims(hi3, "gemm resid3 hi plane")
            if rows < d.M or cols != d.N or hi3.t.dtype != torch.bfloat16:
                raise UiaError(f"gemm resid3 hi plane (K-blocked {tuple(hi3.t.shape)}) does not hold the [{d.M}, {d.N}] bf16 residual")
            d.residT, d.ldrT, d.residT_kb_rows = _p(hi3.t), d.N, plane
        else:
            if hi3.dtype != torch.bfloat16 or hi3.dim() != 2 or hi3.shape[0] < d.M or hi3.shape[1] < d.N:
                raise UiaError(f"gemm resid3 hi plane {tuple(hi3.shape)} {hi3.dtype} does not hold the [{d.M}, {d.N}] bf16 residual")
            d.residT, d.ldrT = _p(hi3), _rowmajor(hi3, "resid3 hi plane")
        if is_kb(lo3):                                   # low bytes in 64-column blocks: KBlocked over an int8 [N/64, rows, 64] tensor
            rows, cols, plane = _kb_dims(lo3, "gemm resid3 low bytes")
            if rows < d.M or cols != d.N or lo3.t.dtype != torch.int8:
                raise UiaError(f"gemm resid3 low bytes (blocked {tuple(lo3.t.shape)}) do not hold the [{d.M}, {d.N}] int8 plane")
            d.resid_lo8, d.ld_resid_lo, d.resid_lo_kb_rows = _p(lo3.t), d.N, plane
        else:
            if lo3.dtype != torch.int8 or lo3.dim() != 2 or lo3.shape[0] < d.M or lo3.shape[1] < d.N:
                raise UiaError(f"gemm resid3 low bytes {tuple(lo3.shape)} {lo3.dtype} do not hold the [{d.M}, {d.N}] int8 plane")
            d.resid_lo8, d.ld_resid_lo = _p(lo3), _rowmajor(lo3, "resid3 low bytes")
    if out_lo is not None:
        if out_t is None:
            raise UiaError("gemm out_lo needs out_t (the hi plane)")
        if is_kb(out_lo):
            rows, cols, plane = _kb_dims(out_lo, "gemm out_lo")
            if rows < d.M or cols != d.N or out_lo.t.dtype != torch.int8:
                raise UiaError(f"gemm out_lo (blocked {tuple(out_lo.t.shape)}) does not hold the [{d.M}, {d.N}] int8 plane")
            d.out_lo8, d.ld_out_lo, d.out_lo_kb_rows = _p(out_lo.t), d.N, plane
        else:
            if out_lo.dtype != torch.int8 or out_lo.dim() != 2 or out_lo.shape[0] < d.M or out_lo.shape[1] < d.N:
                raise UiaError(f"gemm out_lo must be an int8 [{d.M}, {d.N}] plane")
            d.out_lo8, d.ld_out_lo = _p(out_lo), _rowmajor(out_lo, "out_lo")
    if is_kb(out_t):
        d.outT, d.ldo, d.outT_kb_rows = _p(out_t.t), d.N, _kb_dims(out_t, "gemm out_t")[2]
    elif out_t is not None:
        d.outT, d.ldo = _p(out_t), _rowmajor(out_t, "out_t")
    if out32 is not None:
        d.out32, d.ldo32 = _p(out32), _rowmajor(out32, "out32")
    if TILE_GROUP and d.N in TILE_GROUP and (tile_cfg >> 8) == 0 and base_cfg in RING_CFGS:
        tile_cfg |= int(TILE_GROUP[d.N]) << 8
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().uia_gemm(_stream(), _code(a.dtype), C.byref(d), tile_cfg), "uia_gemm")
        e1.record()
        esz = a.element_size()                  # algorithmic HBM bytes of this launch: every operand and result once
        nbytes = esz * (d.M * d.K + d.N * d.K) + d.M * d.N * (esz * sum(t is not None for t in (aux_in, aux_out, resid_t, out_t))
                                                             + 4 * sum(t is not None for t in (resid, out32)) + 3 * (resid3 is not None) + 1 * (out_lo is not None))
        mask = epi_mask_of(d)
        if not (tile_cfg & 255) and base_cfg != 23:
            base_cfg = auto_tile_cfg(d.M, d.N, d.K, esz, mask)
        GEMM_PROFILE.append((e0, e1, d.M, d.N, d.K, a.dtype, base_cfg, nbytes, mask))
        return
    if _TRACE_GENERIC is not None:             # UIA_TRACE_GENERIC=1: which large launches land on the run-time epilogue (a mask worth a compile-time instantiation?)
        mask = epi_mask_of(d)
        if mask not in _SPECIALISED and d.M > 2048:
            _TRACE_GENERIC[(mask, d.M, d.N, d.K, base_cfg)] = _TRACE_GENERIC.get((mask, d.M, d.N, d.K, base_cfg), 0) + 1
    check(lib().uia_gemm(_stream(), _code(a.dtype), C.byref(d), tile_cfg), "uia_gemm")


def three_byte_to_float(hi, lo):
    """The fp32 values of a three-byte tensor: hi = bf16 plane (row-major or KBlocked), lo = int8 plane; float bits = (hi_bits << 16) + (lo << 8)."""
    h = hi.t.permute(1, 0, 2).reshape(hi.rows, hi.cols) if is_kb(hi) else hi
    lo = lo.t.permute(1, 0, 2).reshape(lo.rows, lo.cols) if is_kb(lo) else lo
    h = h[:lo.shape[0], :lo.shape[1]]
    bits = (h.contiguous().view(torch.int16).to(torch.int32) << 16) + (lo.to(torch.int32) << 8)
    return bits.view(torch.float32)


def float_to_three_byte(x):
    """(hi bf16, lo int8) of fp32 `x`, the arithmetic of the GEMM epilogue: hi = round-to-nearest bf16, lo = ((bits + 0x80) >> 8) - (hi_bits << 8) clamped to ±127."""
    hi = x.bfloat16()
    xb = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    hb = hi.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    d = ((xb + 0x80) >> 8) - (hb << 8)
    return hi, d.clamp(-127, 127).to(torch.int8)


def wgrad(a, b, dw, dbias=None, alpha=1.0, drop=None):
    """dw[I,J] += alpha * a.T @ b   (a [M,I], b [M,J]); dbias[I] += a.sum(0).  dw/dbias fp32, pre-zeroed or accumulating.
    dw may be SMALLER than [I, J] (a LoRA factor's own [out, r] / [r, in] gradient under 64-padded operands): only its extent is accumulated."""
    lda, ldb = _rowmajor(a, "a"), _rowmajor(b, "b")
    assert a.dtype == b.dtype and a.shape[0] == b.shape[0] and dw.dtype == torch.float32 and dw.is_contiguous() and dw.dim() == 2
    I, J = a.shape[1], b.shape[1]
    if drop is not None:
        # drop = (p, seed): b holds the UN-dropped rows of a [M, J] tensor (row stride ldb); the mask the forward drew for it is regenerated in the staging
        p, seed = drop
        if dbias is not None or a.dtype != torch.bfloat16 or ldb != J:
            raise UiaError("wgrad(drop=...): bf16, no bias, and b must be the whole dropped tensor (contiguous rows)")
        check(lib().uia_wgrad_drop(_stream(), _code(a.dtype), a.shape[0], I, J, _p(a), lda, _p(b), ldb, alpha, _p(dw), dw.shape[1], dw.shape[0], dw.shape[1],
                                   float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, J, 0), "uia_wgrad_drop")
        return
    if tuple(dw.shape) == (I, J):
        check(lib().uia_wgrad(_stream(), _code(a.dtype), a.shape[0], I, J, _p(a), lda, _p(b), ldb, alpha, _p(dw), _p(dbias)), "uia_wgrad")
        return
    assert dw.shape[0] <= I and dw.shape[1] <= J and (dbias is None or dbias.numel() >= dw.shape[0])
    check(lib().uia_wgrad_ex(_stream(), _code(a.dtype), a.shape[0], I, J, _p(a), lda, _p(b), ldb, alpha, _p(dw), dw.shape[1], dw.shape[0], dw.shape[1],
                             _p(dbias)), "uia_wgrad_ex")


LORA_WGRAD_GROUP = True   # LoRA attention block: the dB (and the dA) weight gradients of q, k and v in ONE launch each (uia_wgrad_group) instead of three; A/B: bench.py --no-lora-wgrad-group


def wgrad_group(a_list, b_list, dw_list, dbias_list=None, alpha=1.0, drop=None):
    """dw_list[g][I', J'] += alpha * a_list[g].T @ b_list[g] for up to four problems of one shape in one launch (uia_wgrad_group); dbias_list[g] += a_list[g].sum(0) where given.
    drop = (p, seeds): b_list[g] holds UN-dropped rows of a contiguous [M, J] tensor and problem g's mask is regenerated from seeds[g] (as wgrad(drop=...) does)."""
    n = len(a_list)
    a0, b0, w0 = a_list[0], b_list[0], dw_list[0]
    if not (1 <= n <= 4 and len(b_list) == n and len(dw_list) == n and a0.dtype == torch.bfloat16):
        raise UiaError("wgrad_group: 1..4 bf16 problems")
    lda, ldb = _rowmajor(a0, "a"), _rowmajor(b0, "b")
    M, I, J = a0.shape[0], a0.shape[1], b0.shape[1]
    d = WgradGroupDesc()
    d.n, d.M, d.I, d.J, d.lda, d.ldb, d.ldw = n, M, I, J, lda, ldb, w0.shape[1]
    d.i_valid, d.j_valid, d.alpha = w0.shape[0], w0.shape[1], alpha
    for g in range(n):
        a, b, w = a_list[g], b_list[g], dw_list[g]
        if (a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or tuple(a.shape) != (M, I) or tuple(b.shape) != (M, J) or _rowmajor(a, "a") != lda or _rowmajor(b, "b") != ldb
                or w.dtype != torch.float32 or not w.is_contiguous() or w.shape != w0.shape or w.shape[0] > I or w.shape[1] > J):
            raise UiaError(f"wgrad_group: problem {g} does not have the shape / strides / dtypes of problem 0")
        d.A[g], d.B[g], d.dW[g] = a.data_ptr(), b.data_ptr(), w.data_ptr()
        db = dbias_list[g] if dbias_list is not None else None
        if db is not None:
            assert db.dtype == torch.float32 and db.numel() >= w.shape[0]
            d.dbias_A[g] = db.data_ptr()
    if drop is not None:
        p_, seeds = drop
        if ldb != J or (dbias_list is not None and any(x is not None for x in dbias_list)):
            raise UiaError("wgrad_group(drop=...): no bias, and each b must be the whole dropped tensor (contiguous rows)")
        d.drop_p, d.drop_ld, d.drop_col0 = float(p_), J, 0
        for g in range(n):
            d.drop_seed[g] = int(seeds[g]) & 0xFFFFFFFFFFFFFFFF
    check(lib().uia_wgrad_group(_stream(), _code(a0.dtype), C.byref(d)), "uia_wgrad_group")


def _attn_desc(q, k, v, out, lse, B, H, L, mask, keylen, scale, dh=None):
    d = AttnDesc()
    for t in (q, k, v):
        if t.stride(-1) != 1 or t.dtype != q.dtype:
            raise UiaError("attention operands must be unit-stride in the last dim and share a dtype")
    # the head dimension is what the operand views say it is: q holds H * dh columns (round 5: callers that left the old default of 64 with a 64-wide, two-head
    # tower — the reference's own toy geometry, tests/golden/openai_clip_base.npz — had head 1 read and WRITE 64 columns past its rows)
    width = q.shape[-1]
    if width % H != 0 or (dh is not None and dh * H != width) or k.shape[-1] != width or v.shape[-1] != width:
        raise UiaError(f"attention: q / k / v are {q.shape[-1]} / {k.shape[-1]} / {v.shape[-1]} columns wide for {H} heads" + (f" of dim {dh}" if dh is not None else ""))
    dh = width // H
    if dh not in (16, 32, 64):
        raise UiaError(f"attention: head dim {dh} unsupported (16, 32 or 64): {width} columns over {H} heads")
    if not is_kb(out) and out.shape[-1] != width:
        raise UiaError(f"attention: out is {out.shape[-1]} columns wide, q / k / v {width}")
    d.q, d.k, d.v, d.ld_qkv = _p(q), _p(k), _p(v), q.stride(-2)
    assert k.stride(-2) == q.stride(-2) == v.stride(-2)
    if is_kb(out):                                     # K-blocked [B*L, H*dh] output (bf16, dh = 64): A operand of the output projection
        rows, cols, d.out_kb_rows = _kb_dims(out, "attention out")
        if rows < B * L or cols != H * dh or dh != 64 or q.dtype != torch.bfloat16:
            raise UiaError(f"attention out (K-blocked {tuple(out.t.shape)}) must hold [{B * L}, {H * dh}] bf16 with head dim 64")
        d.out, d.ldo = _p(out.t), cols
    else:
        d.out, d.ldo = _p(out), out.stride(-2)
    d.lse = _p(lse)
    d.keylen = _p(keylen)
    d.B, d.H, d.L, d.dh = B, H, L, dh
    d.mask_kind = {"none": 0, None: 0, "causal": 1, "keypad": 2}[mask]
    d.scale = scale if scale is not None else dh ** -0.5
    return d


def attn_fwd(q, k, v, out, B, H, L, lse=None, mask=None, keylen=None, scale=None, dh=None, cu_seqlens=None):
    """q,k,v: views whose element (b,l,h,d) is at base[(b*L+l)*ld + h*dh + d] (e.g. slices of the fused qkv).
    cu_seqlens (int32 [B+1], forward only): packed sequences — sequence b is rows cu[b]..cu[b+1]-1, L = the longest one."""
    d = _attn_desc(q, k, v, out, lse, B, H, L, mask, keylen, scale, dh)
    if cu_seqlens is not None:
        assert cu_seqlens.dtype == torch.int32 and cu_seqlens.is_contiguous() and cu_seqlens.numel() == B + 1
        d.cu_seqlens = _p(cu_seqlens)
    check(lib().uia_attn_fwd(_stream(), _code(q.dtype), C.byref(d)), "uia_attn_fwd")


ATTN_BWD_CFG = 0     # uia_attn_bwd_cfg's kernel configuration (0 = the library's choice); tools and tests switch it for A/B runs


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, L, mask=None, keylen=None, scale=None, dh=None, cfg=None):
    d = _attn_desc(q, k, v, out, lse, B, H, L, mask, keylen, scale, dh)
    dh = d.dh
    d.dout, d.lddo = _p(dout), dout.stride(-2)
    if is_kb(dq):                                      # the fused [B*L, 3*H*dh] gradient, K-blocked (dk, dv are then None): A operand of the QKV dgrad
        rows, cols, d.dqkv_kb_rows = _kb_dims(dq, "attention dqkv")
        if rows < B * L or cols != 3 * H * dh or dh != 64 or q.dtype != torch.bfloat16 or dk is not None or dv is not None:
            raise UiaError(f"attention dqkv (K-blocked {tuple(dq.t.shape)}) must hold [{B * L}, {3 * H * dh}] bf16 with head dim 64, dk = dv = None")
        part = (H * dh // kb_group(q.dtype)) * d.dqkv_kb_rows * kb_group(q.dtype) * dq.t.element_size()      # bytes from one part's first column block to the next
        d.dq, d.dk, d.dv, d.ld_dqkv = dq.t.data_ptr(), dq.t.data_ptr() + part, dq.t.data_ptr() + 2 * part, cols
    else:
        d.dq, d.dk, d.dv, d.ld_dqkv = _p(dq), _p(dk), _p(dv), dq.stride(-2)
        assert dk.stride(-2) == dq.stride(-2) == dv.stride(-2)
    cfg = ATTN_BWD_CFG if cfg is None else cfg
    if cfg:
        check(lib().uia_attn_bwd_cfg(_stream(), _code(q.dtype), C.byref(d), int(cfg)), "uia_attn_bwd_cfg")
    else:
        check(lib().uia_attn_bwd(_stream(), _code(q.dtype), C.byref(d)), "uia_attn_bwd")


def layernorm_fwd(x, gamma, beta, eps, y_t=None, y32=None, rows=None, ldx=None, stats=None):
    """x fp32 [rows, D] (row stride ldx); y_t (bf16|fp32) and/or y32 compact [rows, D]; stats fp32 [rows, 2] receives (mean, rstd)."""
    D = gamma.numel()
    rows = x.numel() // D if rows is None else rows
    ldx = D if ldx is None else ldx
    dt = _code(y_t.dtype) if y_t is not None else _lib.F32
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.numel() >= 2 * rows
        check(lib().uia_layernorm_fwd_stats(_stream(), dt, rows, D, ldx, _p(x), _p(gamma), _p(beta), eps, _p(y_t), _p(y32), _p(stats)), "uia_layernorm_fwd_stats")
        return
    check(lib().uia_layernorm_fwd(_stream(), dt, rows, D, ldx, _p(x), _p(gamma), _p(beta), eps, _p(y_t), _p(y32)), "uia_layernorm_fwd")


def layernorm_bwd(dy, x, gamma, eps, dres=None, dx32=None, dx_t=None, rows=None, ldx=None, dx_lo=None):
    """dx = dres + LN'(dy) on rows of x.  x and dres may be three-byte tensors, passed as (hi, lo) tuples (hi bf16 row-major — x's hi plane may be
    KBlocked — lo int8 [M, D]); dx_lo = int8 [M, D] makes the result a three-byte tensor whose hi plane is dx_t (dx32 may then be None)."""
    D = gamma.numel()
    rows = dy.numel() // D if rows is None else rows
    ldx = D if ldx is None else ldx
    if isinstance(x, tuple) or isinstance(dres, tuple) or dx_lo is not None:
        x_hi, x_lo = x if isinstance(x, tuple) else (None, None)
        r_hi, r_lo = dres if isinstance(dres, tuple) else (None, None)
        x_kb = r_kb = 0
        if is_kb(x_hi):
            kr, kc, x_kb = _kb_dims(x_hi, "layernorm_bwd x hi plane")
            if kr < rows or kc != D:
                raise UiaError(f"layernorm_bwd: K-blocked x hi plane {tuple(x_hi.t.shape)} does not hold [{rows}, {D}]")
            x_hi = x_hi.t
        if is_kb(r_hi):
            kr, kc, r_kb = _kb_dims(r_hi, "layernorm_bwd dres hi plane")
            if kr < rows or kc != D:
                raise UiaError(f"layernorm_bwd: K-blocked dres hi plane {tuple(r_hi.t.shape)} does not hold [{rows}, {D}]")
            r_hi = r_hi.t
        for name, t, dt_ in (("x hi", x_hi, torch.bfloat16), ("x lo", x_lo, torch.int8), ("dres hi", r_hi, torch.bfloat16), ("dres lo", r_lo, torch.int8), ("dx_lo", dx_lo, torch.int8)):
            if t is not None and (t.dtype != dt_ or not t.is_contiguous() or t.numel() < rows * D):
                raise UiaError(f"layernorm_bwd: {name} plane must be a contiguous {dt_} tensor of at least [{rows}, {D}], got {tuple(t.shape)} {t.dtype}")
        check(lib().uia_layernorm_bwd3(_stream(), _code(dy.dtype), rows, D, ldx, _p(dy), None if isinstance(x, tuple) else _p(x), _p(x_hi), _p(x_lo), x_kb, _p(gamma), eps,
                                       None if isinstance(dres, tuple) else _p(dres), _p(r_hi), _p(r_lo), r_kb, _p(dx32), _p(dx_t), _p(dx_lo)), "uia_layernorm_bwd3")
        return
    check(lib().uia_layernorm_bwd(_stream(), _code(dy.dtype), rows, D, ldx, _p(dy), _p(x), _p(gamma), eps, _p(dres), _p(dx32), _p(dx_t)), "uia_layernorm_bwd")


def cast(src, dst, scale=1.0):
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    check(lib().uia_cast(_stream(), _code(dst.dtype), src.numel(), _p(src), _p(dst), scale), "uia_cast")


def transpose_cast(src, dst):
    assert src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous() and dst.is_contiguous()
    assert tuple(dst.shape) == (src.shape[1], src.shape[0])
    check(lib().uia_transpose_cast(_stream(), _code(dst.dtype), src.shape[0], src.shape[1], _p(src), _p(dst)), "uia_transpose_cast")


_PIN_RING, _PIN_SLOTS, _PIN_SLOT = {}, 512, 4096


def pack_table(entries, device):
    """Device-resident uia_pack_desc table for `entries` = [(src fp32 [R, C], row, row_kb, tr, tr_kb[, (rows_pad, cols_pad, scale)])] (None = form
    not wanted).  With the optional sixth element the destinations hold rows_pad x cols_pad elements (zero padding kept by the caller: the
    launch writes only the source's elements).  Returns (table tensor, n, max_elems); keep the tensors of `entries` alive as long as the table is used."""
    arr = (PackDesc * len(entries))()
    max_elems = 0
    for d, ent in zip(arr, entries):
        src, row, row_kb, tr, tr_kb = ent[:5]
        assert src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous()
        R, Cc = src.shape
        RP, CP, scale = ent[5] if len(ent) > 5 and ent[5] is not None else (R, Cc, 1.0)
        assert RP >= R and CP >= Cc
        d.src, d.rows, d.cols, d.rows_pad, d.cols_pad, d.scale = _p(src), R, Cc, RP, CP, float(scale)
        for name, t in (("row", row), ("row_kb", row_kb), ("tr", tr), ("tr_kb", tr_kb)):
            if isinstance(t, RawDest):
                assert t.view.device == src.device
                setattr(d, name, t.view.data_ptr())
            elif t is not None:
                assert t.is_contiguous() and t.numel() == RP * CP and t.device == src.device
                setattr(d, name, _p(t))
        max_elems = max(max_elems, R * Cc)
    nbytes = C.sizeof(arr)
    dev = torch.device(device)
    if dev.type == "cuda" and nbytes <= _PIN_SLOT:
        # through a ring of PINNED host slots and an asynchronous copy: a pageable host-to-device copy waits for the stream, and a step that builds derived weights
        # (CLIPSeg's decoder: concatenated q | k | v, conv kernels as GEMM weights — new tensors every step) paid that wait six times (1.4 ms of a host-bound 10 ms step).
        # 512 slots: the host cannot run 512 pack launches ahead of the device.
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        ring = _PIN_RING.get(key)
        if ring is None:
            ring = _PIN_RING[key] = [torch.empty(_PIN_SLOTS, _PIN_SLOT, dtype=torch.uint8).pin_memory(), 0, [None] * _PIN_SLOTS]
        i = ring[1]
        slot = ring[0][i]
        ring[1] = (i + 1) % _PIN_SLOTS
        if ring[2][i] is not None:
            ring[2][i].synchronize()        # the copy that last read this slot (512 pack launches ago: long done; a wait only if the host ran THAT far ahead — ADVICE r04)
        C.memmove(slot.data_ptr(), C.addressof(arr), nbytes)
        table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        table.copy_(slot[:nbytes], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring[2][i] = ev
        return table, len(entries), max_elems
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
    return raw.to(device), len(entries), max_elems


def pack_weights(table, n, max_elems, dtype):
    check(lib().uia_pack_weights(_stream(), _code(dtype), n, _p(table), max_elems), "uia_pack_weights")


def im2col(img, out, patch):
    """out [B*gh*gw, ld]: ld == C*patch^2 with patch % 4 == 0 takes the vectorised kernel, anything else the padded one
    (columns beyond C*patch^2 are zero-filled, e.g. ViT-L/14: 588 -> 640 for the GEMM's K granule)."""
    B, Cc, H, W = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous() and out.is_contiguous()
    if patch % 4 == 0 and out.shape[1] == Cc * patch * patch and W % 4 == 0:
        check(lib().uia_im2col(_stream(), _code(out.dtype), B, Cc, H, W, patch, _p(img), _p(out)), "uia_im2col")
    else:
        check(lib().uia_im2col_padded(_stream(), _code(out.dtype), B, Cc, H, W, patch, _p(img), _p(out), out.shape[1]), "uia_im2col_padded")


def fill_cls(x, cls, pos0):
    B, N, D = x.shape
    check(lib().uia_fill_cls(_stream(), B, N, D, _p(cls), _p(pos0), _p(x)), "uia_fill_cls")


def embed(ids, table, pos, type0, out):
    rows, L = ids.numel(), ids.shape[-1]
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    check(lib().uia_embed(_stream(), rows, L, table.shape[1], table.shape[0], pos.shape[0], _p(ids), _p(table), _p(pos), _p(type0), _p(out)), "uia_embed")


def embed_packed(ids, pos_idx, table, pos, type0, out):
    """out[r] = table[ids[r]] + pos[pos_idx[r]] + type0 for the packed (valid) tokens r."""
    assert ids.dtype == pos_idx.dtype == torch.int64 and ids.is_contiguous() and pos_idx.is_contiguous() and ids.numel() == pos_idx.numel() == out.shape[0]
    check(lib().uia_embed_packed(_stream(), ids.numel(), out.shape[1], table.shape[0], pos.shape[0], _p(ids), _p(pos_idx), _p(table), _p(pos), _p(type0), _p(out)),
          "uia_embed_packed")


def embed_bwd(ids, dx, dtable, pad_id=-1):
    """dtable[ids[r]] += dx[r] for every row r with ids[r] != pad_id (dtable fp32, caller-zeroed or accumulating)."""
    assert ids.dtype == torch.int64 and ids.is_contiguous() and dx.dtype == dtable.dtype == torch.float32 and dx.is_contiguous() and dtable.is_contiguous()
    check(lib().uia_embed_bwd(_stream(), ids.numel(), dx.shape[-1], dtable.shape[0], _p(ids), _p(dx), _p(dtable), pad_id), "uia_embed_bwd")


def gather_rows(src, idx, dst):
    assert idx.dtype == torch.int64
    check(lib().uia_gather_rows(_stream(), idx.numel(), src.shape[-1], _p(src), _p(idx), _p(dst)), "uia_gather_rows")


# ------------------------------------------------------------------------------------------- Mona
_SPATIAL_KEYS = ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "conv3_b", "proj_w", "proj_b", "freq", "ne1_w", "ne1_b", "ne3_w", "ne3_b")


def mona_pre_fwd(x, norm_w, norm_b, gamma, gammax, u, eps=1e-5, proj1=None):
    """proj1 = (w1 [64, D] T row-major, b1 fp32 [64] or None, t [M, 64] T): project1 runs inside the launch (uia_mona_pre_fwd_t: bf16, D = 768)."""
    D = gamma.numel()
    if proj1 is not None:
        w1, b1, t = proj1
        M = x.numel() // D
        if u.dtype != torch.bfloat16 or w1.dtype != u.dtype or t.dtype != u.dtype or tuple(w1.shape) != (64, D) or tuple(t.shape) != (M, 64):
            raise UiaError(f"mona_pre_fwd proj1: w1 {tuple(w1.shape)} / t {tuple(t.shape)} must be bf16 [64, {D}] / [{M}, 64]")
        check(lib().uia_mona_pre_fwd_t(_stream(), _code(u.dtype), M, D, _p(x), _p(norm_w), _p(norm_b), _p(gamma), _p(gammax), eps, _p(u), _p(w1), _rowmajor(w1, "w1"),
                                       _p(b1), _p(t), _rowmajor(t, "t")), "uia_mona_pre_fwd_t")
        return
    check(lib().uia_mona_pre_fwd(_stream(), _code(u.dtype), x.numel() // D, D, _p(x), _p(norm_w), _p(norm_b), _p(gamma), _p(gammax), eps, _p(u)), "uia_mona_pre_fwd")


def mona_pre_bwd_du_ok(M, D, bott, dt):
    """Shapes uia_mona_pre_bwd_du takes: bf16, bottleneck 64, D a multiple of 64 up to 768 (the row kernel's three-float4-per-lane instantiation)."""
    return MONA_PRE_BWD_DU and dt == torch.bfloat16 and bott == 64 and D % 64 == 0 and 512 < D <= 768


def mona_pre_bwd(du, x, dy, norm_w, norm_b, gamma, gammax, dx32, dx_t, g_gamma, g_gammax, g_norm_w, g_norm_b, eps=1e-5, dt_w1t=None, dx_lo=None):
    """dt_w1t = (dt [M, 64], W1ᵀ [D, 64]) instead of du: project1's data gradient du = dt·W1 is computed inside the launch (uia_mona_pre_bwd_du).
    Three-byte residual gradients (uia_mona_pre_bwd_du3, with dt_w1t only): dy = (hi bf16 [M, D] row-major, lo int8 [M, D]) and dx_lo = int8 [M, D] — the result
    is then (dx_t, dx_lo) and dx32 must be None."""
    D = gamma.numel()
    M = x.numel() // D
    ws = torch.empty(lib().uia_mona_pre_bwd_workspace_bytes(M, D) // 4, device=x.device, dtype=torch.float32)
    kb_rows = 0
    if is_kb(dx_t):                                    # K-blocked T copy of dx (read by a ring GEMM only)
        rows, cols, kb_rows = _kb_dims(dx_t, "mona_pre_bwd dx_t")
        if rows != M or cols != D:
            raise UiaError(f"mona_pre_bwd dx_t (K-blocked {tuple(dx_t.t.shape)}) does not hold [{M}, {D}]")
        dx_t = dx_t.t
    if dt_w1t is not None:
        dtt, w1t = dt_w1t
        if dtt.dtype != torch.bfloat16 or w1t.dtype != torch.bfloat16 or tuple(dtt.shape) != (M, 64) or tuple(w1t.shape) != (D, 64):
            raise UiaError(f"mona_pre_bwd: dt {tuple(dtt.shape)} {dtt.dtype} / W1ᵀ {tuple(w1t.shape)} {w1t.dtype} must be bf16 [{M}, 64] / [{D}, 64]")
        if isinstance(dy, tuple) or dx_lo is not None:
            if not (isinstance(dy, tuple) and dx_lo is not None and dx32 is None and dx_t is not None):
                raise UiaError("mona_pre_bwd: three-byte residual gradients need dy = (hi, lo), dx_lo and dx_t, and no dx32")
            hi, lo = dy
            for name, t_, dt_ in (("dy hi", hi, torch.bfloat16), ("dy lo", lo, torch.int8), ("dx_lo", dx_lo, torch.int8), ("dx_t", dx_t, torch.bfloat16)):
                if t_.dtype != dt_ or not t_.is_contiguous() or t_.numel() < M * D:
                    raise UiaError(f"mona_pre_bwd: {name} plane must be a contiguous {dt_} tensor of at least [{M}, {D}], got {tuple(t_.shape)} {t_.dtype}")
            check(lib().uia_mona_pre_bwd_du3(_stream(), _code(dtt.dtype), M, D, _p(dtt), _rowmajor(dtt, "dt"), _p(w1t), _rowmajor(w1t, "w1t"), _p(x), _p(hi), _p(lo), _p(norm_w),
                                             _p(norm_b), _p(gamma), _p(gammax), eps, _p(dx_t), _p(dx_lo), _p(g_gamma), _p(g_gammax), _p(g_norm_w), _p(g_norm_b), _p(ws),
                                             kb_rows), "uia_mona_pre_bwd_du3")
            return
        check(lib().uia_mona_pre_bwd_du(_stream(), _code(dtt.dtype), M, D, _p(dtt), _rowmajor(dtt, "dt"), _p(w1t), _rowmajor(w1t, "w1t"), _p(x), _p(dy), _p(norm_w),
                                        _p(norm_b), _p(gamma), _p(gammax), eps, _p(dx32), _p(dx_t), _p(g_gamma), _p(g_gammax), _p(g_norm_w), _p(g_norm_b), _p(ws),
                                        kb_rows), "uia_mona_pre_bwd_du")
        return
    check(lib().uia_mona_pre_bwd(_stream(), _code(du.dtype), M, D, _p(du), _p(x), _p(dy), _p(norm_w), _p(norm_b), _p(gamma), _p(gammax),
                                 eps, _p(dx32), _p(dx_t), _p(g_gamma), _p(g_gammax), _p(g_norm_w), _p(g_norm_b), _p(ws), kb_rows), "uia_mona_pre_bwd")


def _spatial_desc(variant, B, h, w, t, params, p_drop, seed, keep_mask):
    d = MonaSpatialDesc()
    d.variant = _lib.MONA_VARIANTS[variant]
    d.B, d.h, d.w, d.bott = B, h, w, t.shape[-1]
    d.t = _p(t)
    for k in _SPATIAL_KEYS:
        v = params.get(k)
        if v is not None:
            assert v.dtype == torch.float32 and v.is_contiguous()
            setattr(d, k, _p(v))
    d.p_drop, d.seed = float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF
    if keep_mask is not None:
        assert keep_mask.dtype == torch.uint8 and keep_mask.is_contiguous()
        d.keep_mask = _p(keep_mask)
    return d


# The whole adapter forward in one launch (csrc/mona_fused.hip) where uia_mona_fused_supported says so.  Built, parity-tested and measured in
# round 3 (tools/mff_variants.sh, profiles/r03_c_mona_fused_phases.txt): 199 us per ViT-B/16 layer with the u stash the backward's weight
# gradient needs (181 without) against 179 us for the four unfused launches — its third phase IS the K = 64 projection with the fp32 residual
# (387 MB, ~100 us either way) and the first two save 16 us of the 98 they replace.  Opt-in until the backward recomputes u.
MONA_FUSED = False


def mona_fused_ok(dt, D, h, w, bott):
    return MONA_FUSED and dt == torch.bfloat16 and bool(lib().uia_mona_fused_supported(_code(dt), D, h, w, bott))


def mona_fused_fwd(variant, B, h, w, x, norm_w, norm_b, gamma, gammax, w1, b1, w2, b2, params, y32, y_t=None, rowsum=None, u_out=None, t_out=None, d_out=None,
                   p_drop=0.0, seed=0, keep_mask=None, eps=1e-5):
    """y32 = x + project2(drop(gelu(spatial(project1(LN(x)·gamma + x·gammax)))))  for B images of 1 + h·w tokens, one launch
    (include/uia_hip.h, uia_mona_fused_desc).  x fp32 [B, 1+h·w, D]; w1 [64, D] / w2 [D, 64] in the compute dtype; params as mona_spatial_fwd.
    y_t: optional T copy of y (tensor or KBlocked); rowsum: int64 [M, 2] receiving the rows' (Σ, Σ²); u_out / t_out / d_out: optional stashes."""
    D = x.shape[-1]
    q = MonaFusedDesc()
    sp = _spatial_desc(variant, B, h, w, w1.new_empty(0, w1.shape[0]) if t_out is None else t_out, params, p_drop, seed, keep_mask)
    sp.t = None
    if d_out is not None:
        sp.d = _p(d_out)
    q.sp = sp
    q.D, q.eps = D, eps
    assert x.dtype == torch.float32 and x.is_contiguous() and y32.dtype == torch.float32 and y32.is_contiguous() and y32.numel() == x.numel()
    assert tuple(w1.shape) == (sp.bott, D) and tuple(w2.shape) == (D, sp.bott) and w1.is_contiguous() and w2.is_contiguous() and w1.dtype == w2.dtype
    q.x, q.norm_w, q.norm_b, q.gamma, q.gammax = _p(x), _p(norm_w), _p(norm_b), _p(gamma), _p(gammax)
    q.w1, q.b1, q.w2, q.b2, q.y32 = _p(w1), _p(b1), _p(w2), _p(b2), _p(y32)
    M = B * (1 + h * w)
    if is_kb(y_t):
        rows, cols, plane = _kb_dims(y_t, "mona_fused y_t")
        if rows < M or cols != D:
            raise UiaError(f"mona_fused y_t (K-blocked {tuple(y_t.t.shape)}) does not hold [{M}, {D}]")
        q.yT, q.yT_kb_rows = _p(y_t.t), plane
    elif y_t is not None:
        assert y_t.dtype == w1.dtype and y_t.is_contiguous() and y_t.numel() == M * D
        q.yT = _p(y_t)
    if rowsum is not None:
        assert rowsum.dtype == torch.int64 and rowsum.is_contiguous() and rowsum.numel() >= 2 * M
        q.rowsum_out, q.ln_flag = _p(rowsum), _p(ln_flag(x.device))
    for name, t, width in (("u_out", u_out, D), ("t_out", t_out, sp.bott)):
        if t is not None:
            assert t.dtype == w1.dtype and t.is_contiguous() and t.numel() == M * width
            setattr(q, name, _p(t))
    check(lib().uia_mona_fused_fwd(_stream(), _code(w1.dtype), C.byref(q)), "uia_mona_fused_fwd")


def mona_spatial_fwd(variant, B, h, w, t, params, d_out, p_drop=0.0, seed=0, keep_mask=None):
    """params: dict with keys of _SPATIAL_KEYS (fp32, contiguous; absent ones for variants without them)."""
    d = _spatial_desc(variant, B, h, w, t, params, p_drop, seed, keep_mask)
    d.d = _p(d_out)
    check(lib().uia_mona_spatial_fwd(_stream(), _code(t.dtype), C.byref(d)), "uia_mona_spatial_fwd")


def mona_spatial_bwd(variant, B, h, w, t, params, dd, dt, grads, p_drop=0.0, seed=0, keep_mask=None):
    """grads: dict keyed like params with fp32 accumulators (caller-zeroed)."""
    d = _spatial_desc(variant, B, h, w, t, params, p_drop, seed, keep_mask)
    d.dd, d.dt = _p(dd), _p(dt)
    ws = torch.empty(lib().uia_mona_spatial_workspace_bytes(B) // 4, device=t.device, dtype=torch.float32)
    d.ws = _p(ws)
    for k in _SPATIAL_KEYS:
        v = grads.get(k)
        if v is not None:
            setattr(d, "g_" + k, _p(v))
    check(lib().uia_mona_spatial_bwd(_stream(), _code(t.dtype), C.byref(d)), "uia_mona_spatial_bwd")


# ------------------------------------------------------------------------------------------- loss / optimiser / comm
def infonce(img, txt, temperature, grad_scale=1.0, want_grads=True):
    """Returns (loss[1] fp32 device tensor, dimg, dtxt).  img/txt fp32 [B,E] contiguous."""
    assert img.dtype == torch.float32 and txt.dtype == torch.float32 and img.is_contiguous() and txt.is_contiguous()
    B, E = img.shape
    nbytes = lib().uia_infonce_workspace_bytes(B, E)
    ws = torch.empty(nbytes // 4, device=img.device, dtype=torch.float32)
    loss = torch.empty(1, device=img.device, dtype=torch.float32)
    dimg = torch.empty_like(img) if want_grads else None
    dtxt = torch.empty_like(txt) if want_grads else None
    check(lib().uia_infonce_fwd_bwd(_stream(), B, E, _p(img), _p(txt), 1.0 / temperature, grad_scale, _p(loss), _p(dimg), _p(dtxt), _p(ws), nbytes), "uia_infonce_fwd_bwd")
    return loss, dimg, dtxt


def adamw_clip_step(p, g, m, v, lr, betas, eps, weight_decay, max_norm, step, grad_scale, ws2):
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    check(lib().uia_adamw_clip_step(_stream(), p.numel(), _p(p), _p(g), _p(m), _p(v), lr, betas[0], betas[1], eps, weight_decay, max_norm, step, grad_scale, _p(ws2)), "uia_adamw_clip_step")


def grad_accum_guarded(acc, mb, loss, stats, ctl, ok_log=None, log_index=0):
    """acc [n + 4] += mb [n] when the device loss is finite; mb zeroed; acc[n] = this micro-batch's flag (include/uia_hip.h)."""
    n = mb.numel()
    assert acc.dtype == mb.dtype == loss.dtype == stats.dtype == torch.float32 and acc.numel() >= n + 4 and ctl.dtype == torch.int32 and ctl.numel() >= 4
    assert ok_log is None or (ok_log.dtype == torch.uint8 and 0 <= log_index < ok_log.numel())
    check(lib().uia_grad_accum_guarded(_stream(), n, _p(acc), _p(mb), _p(loss), _p(stats), _p(ctl), _p(ok_log), int(log_index)), "uia_grad_accum_guarded")


def adamw_clip_step_guarded(p, acc, m, v, lr, lr_min, t_max, betas, eps, weight_decay, max_norm, grad_scale, skip_scale, ws8, ctl):
    n = p.numel()
    for t in (p, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    assert acc.dtype == torch.float32 and acc.numel() >= n + 4 and ws8.numel() >= 8 and ctl.dtype == torch.int32 and ctl.numel() >= 4
    check(lib().uia_adamw_clip_step_guarded(_stream(), n, _p(p), _p(acc), _p(m), _p(v), lr, lr_min, int(t_max), betas[0], betas[1], eps, weight_decay, max_norm,
                                            grad_scale, skip_scale, _p(ws8), _p(ctl)), "uia_adamw_clip_step_guarded")


def comm_unique_id():
    n = lib().uia_comm_unique_id_bytes()
    buf = C.create_string_buffer(n)
    check(lib().uia_comm_get_unique_id(buf, n), "uia_comm_get_unique_id")
    return bytes(buf.raw)


def comm_init(rank, world, uid):
    check(lib().uia_comm_init(rank, world, C.c_char_p(uid), len(uid)), "uia_comm_init")


def comm_world():
    return lib().uia_comm_world()


def comm_world_initialised():
    return bool(lib().uia_comm_initialised())


def allreduce_sum(buf):
    import os
    try:
        check(lib().uia_allreduce_sum(_stream(), _code(buf.dtype), _p(buf), buf.numel()), "uia_allreduce_sum")
    except UiaError as e:                                       # say WHICH rank: the others are most likely blocked inside the collective
        raise UiaError(f"rank {os.environ.get('RANK', '0')}/{os.environ.get('WORLD_SIZE', '1')}: {e}") from e


def allgather(send, recv):
    """recv [world * n] = concatenation over ranks of send [n] (same dtype, contiguous)."""
    assert send.is_contiguous() and recv.is_contiguous() and send.dtype == recv.dtype and recv.numel() == send.numel() * comm_world()
    check(lib().uia_allgather(_stream(), _code(send.dtype), _p(send), _p(recv), send.numel()), "uia_allgather")


def comm_destroy():
    check(lib().uia_comm_destroy(), "uia_comm_destroy")


def dropout(src, dst, p, seed, accumulate=False):
    assert src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    check(lib().uia_dropout(_stream(), _code(src.dtype), src.numel(), _p(src), _p(dst), p, int(seed) & 0xFFFFFFFFFFFFFFFF, int(accumulate)), "uia_dropout")


def ln_lora_down_ok(D, r, dt):
    """Shapes uia_ln_lora_down takes: bf16, D = 768 or 1024, rank <= 16."""
    return LN_LORA_DOWN and dt == torch.bfloat16 and D in (768, 1024) and 0 < r <= 16


def ln_lora_down(x, gamma, beta, eps, h, a_rows, t_all, drop_p=0.0, seeds=()):
    """h = LayerNorm(x) (T [M, D], written) and t_all[s] = dropout_s(h) @ a_rows[s][:16].T for up to three LoRA wrappers sharing the input, in one launch.
    a_rows: T [>= 16, D] each (rank rows, zero-padded); t_all: T [n, M, 64] contiguous — columns 16..63 are written as zeros."""
    n, M, w = t_all.shape
    D = gamma.numel()
    if not (t_all.is_contiguous() and w == 64 and len(a_rows) == n and tuple(h.shape) == (M, D) and h.is_contiguous() and x.dtype == torch.float32 and x.shape[-1] == D):
        raise UiaError(f"ln_lora_down: x {tuple(x.shape)}, h {tuple(h.shape)}, t_all {tuple(t_all.shape)}")
    if drop_p > 0 and len(seeds) != n:
        raise UiaError("ln_lora_down: one dropout seed per source")
    d = LnLoraDesc()
    d.M, d.D, d.nsrc, d.eps = M, D, n, float(eps)
    d.x, d.ldx = _p(x), _rowmajor(x, "x")
    d.gamma, d.beta, d.h = _p(gamma), _p(beta), _p(h)
    lda = None
    for i, a in enumerate(a_rows):
        if a.dtype != h.dtype or a.dim() != 2 or a.shape[0] < 16 or a.shape[1] != D:
            raise UiaError(f"ln_lora_down: factor {i} is {tuple(a.shape)} {a.dtype}, expected [>= 16, {D}] {h.dtype}")
        l = _rowmajor(a, "a")
        if lda is not None and l != lda:
            raise UiaError("ln_lora_down: the factors must share one leading dimension")
        lda = l
        d.A[i] = _p(a)
    d.lda = lda
    d.T, d.t_stride = _p(t_all), M * 64
    d.drop_p = float(drop_p)
    for i in range(n if drop_p > 0 else 0):
        d.seed[i] = int(seeds[i]) & 0xFFFFFFFFFFFFFFFF
    check(lib().uia_ln_lora_down(_stream(), _code(h.dtype), C.byref(d)), "uia_ln_lora_down")


def lora_rank_update_ok(n_src, N, dt):
    """Shapes uia_lora_rank_update takes: bf16, N a multiple of 256, the sources' [N/4, 64] weight rows inside the LDS."""
    return dt == torch.bfloat16 and 1 <= n_src <= 3 and N % 256 == 0 and n_src * (N // 4) * 144 <= 160 * 1024


def lora_rank_update(q_all, ws, out, alpha, drop_p=0.0, seeds=()):
    """out += sum_i drop_i(alpha * q_all[i] @ ws[i].T) in ONE read-modify-write pass over `out` (uia_lora_rank_update, csrc/lora_rank.hip).
    q_all: T [n, M, 64] (contiguous); ws: n matrices T [N, 64] (A_i transposed, rank zero-padded to 64); out: T [M, N] row-major;
    seeds: one per source when drop_p > 0 — the masks are those uia_dropout / gemm(drop=("acc", p, seed)) draw for an [M, N] tensor."""
    n, M, r = q_all.shape
    N = out.shape[1]
    if not (q_all.is_contiguous() and r == 64 and out.shape[0] == M and len(ws) == n and lora_rank_update_ok(n, N, out.dtype) and q_all.dtype == out.dtype):
        raise UiaError(f"lora_rank_update: q_all {tuple(q_all.shape)} {q_all.dtype}, out {tuple(out.shape)} {out.dtype}, {len(ws)} weights")
    if drop_p > 0 and len(seeds) != n:
        raise UiaError("lora_rank_update: one dropout seed per source")
    d = LoraRankDesc()
    d.M, d.N, d.nsrc, d.alpha = M, N, n, float(alpha)
    d.Q, d.ldq, d.q_stride = _p(q_all), 64, M * 64
    ldw = None
    for i, w in enumerate(ws):
        if w.dtype != out.dtype or tuple(w.shape) != (N, 64):
            raise UiaError(f"lora_rank_update: weight {i} is {tuple(w.shape)} {w.dtype}, expected [{N}, 64] {out.dtype}")
        l = _rowmajor(w, "w")
        if ldw is not None and l != ldw:
            raise UiaError("lora_rank_update: the weights must share one leading dimension")
        ldw = l
        d.W[i] = _p(w)
    d.ldw = ldw
    d.out, d.ldo = _p(out), _rowmajor(out, "out")
    d.drop_p = float(drop_p)
    for i in range(n if drop_p > 0 else 0):
        d.seed[i] = int(seeds[i]) & 0xFFFFFFFFFFFFFFFF
    check(lib().uia_lora_rank_update(_stream(), _code(out.dtype), C.byref(d)), "uia_lora_rank_update")


def colsum(a, out):
    lda = _rowmajor(a, "a")
    assert out.dtype == torch.float32 and out.numel() == a.shape[1]
    check(lib().uia_colsum(_stream(), _code(a.dtype), a.shape[0], a.shape[1], _p(a), lda, _p(out)), "uia_colsum")


# ------------------------------------------------------------------------------------------- CLIPSeg decoder pieces
def layernorm_bwd_affine(dy, x, gamma, eps, dx32, g_gamma, g_beta, dres=None):
    D = gamma.numel()
    check(lib().uia_layernorm_bwd_affine(_stream(), _code(dy.dtype), dy.numel() // D, D, _p(dy), _p(x), _p(gamma), eps, _p(dres), _p(dx32), _p(g_gamma), _p(g_beta)),
          "uia_layernorm_bwd_affine")


def film_fwd(x, mul, add, y):
    B, N, Cc = x.shape
    check(lib().uia_film_fwd(_stream(), B, N, Cc, _p(x), _p(mul), _p(add), _p(y)), "uia_film_fwd")


def film_bwd(dy, x, mul, dx, dmul, dadd):
    B, N, Cc = x.shape
    check(lib().uia_film_bwd(_stream(), B, N, Cc, _p(dy), _p(x), _p(mul), _p(dx), _p(dmul), _p(dadd)), "uia_film_bwd")


def im2col3x3(x, cols, h, w, tok_off=1):
    B, N, Cc = x.shape
    check(lib().uia_im2col3x3(_stream(), _code(cols.dtype), B, h, w, Cc, N, tok_off, _p(x), _p(cols)), "uia_im2col3x3")


def col2im3x3(dcols, dx, h, w, tok_off=1):
    B, N, Cc = dx.shape
    check(lib().uia_col2im3x3(_stream(), _code(dcols.dtype), B, h, w, Cc, N, tok_off, _p(dcols), _p(dx)), "uia_col2im3x3")


def unshuffle(tmp, out, B, h, w, k1, k2, bias=0.0):
    check(lib().uia_unshuffle(_stream(), _code(tmp.dtype), B, h, w, k1, k2, _p(tmp), tmp.stride(0), bias, _p(out)), "uia_unshuffle")


def shuffle(dout, dtmp, B, h, w, k1, k2):
    check(lib().uia_shuffle(_stream(), _code(dtmp.dtype), B, h, w, k1, k2, _p(dout), _p(dtmp), dtmp.stride(0)), "uia_shuffle")


def act_bwd(dy, y, act, out):
    check(lib().uia_act_bwd(_stream(), _code(dy.dtype), dy.numel(), _p(dy), _p(y), _ACT[act], _p(out)), "uia_act_bwd")


# ------------------------------------------------------------------------------------------- FPN task heads
def upsample_bilinear(src, B, C, h, w, H, W, dst, backward=False):
    """forward: src token-major fp32 [B*h*w, ld>=C] -> dst [B,C,H,W]; backward: src = d(dst) [B,C,H,W] -> dst token-major (overwritten)."""
    tok = dst if backward else src
    assert tok.dtype == torch.float32 and tok.dim() == 2 and tok.stride(1) == 1 and tok.shape[0] == B * h * w and tok.shape[1] >= C
    img = src if backward else dst
    assert img.dtype == torch.float32 and img.is_contiguous() and tuple(img.shape) == (B, C, H, W)
    if backward:
        check(lib().uia_upsample_bilinear_bwd(_stream(), B, C, h, w, H, W, _p(src), _p(dst), dst.stride(0)), "uia_upsample_bilinear_bwd")
    else:
        check(lib().uia_upsample_bilinear_fwd(_stream(), B, C, h, w, H, W, _p(src), src.stride(0), _p(dst)), "uia_upsample_bilinear_fwd")


def segment_mean(x, B, n, out, backward=False):
    """forward: x fp32 [B*n, C] -> out [B, C] (mean over each image's n tokens); backward: x = d(out) [B, C] -> out [B*n, C]."""
    tok = out if backward else x
    vec = x if backward else out
    Cc = vec.shape[1]
    assert tok.dtype == vec.dtype == torch.float32 and tok.stride(1) == 1 and vec.is_contiguous() and tok.shape[0] == B * n and tok.shape[1] == Cc
    if backward:
        check(lib().uia_segment_mean_bwd(_stream(), B, n, Cc, _p(x), _p(out), out.stride(0)), "uia_segment_mean_bwd")
    else:
        check(lib().uia_segment_mean_fwd(_stream(), B, n, Cc, _p(x), x.stride(0), _p(out)), "uia_segment_mean_fwd")


def dicece_fwd_bwd(logits, label, smooth_nr=1e-8, smooth_dr=1e-8):
    """logits fp32 [B,C,H,W], label [B,1,H,W] (class indices) -> (loss 0-dim fp32, dlogits fp32 [B,C,H,W])."""
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 4
    B, Cc, H, W = logits.shape
    lab = label.reshape(B, H * W).to(torch.float32).contiguous()
    ws = torch.empty(lib().uia_dicece_workspace_bytes(B) // 4, device=logits.device, dtype=torch.float32)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits)
    check(lib().uia_dicece_fwd_bwd(_stream(), B, Cc, H * W, _p(logits), _p(lab), smooth_nr, smooth_dr, _p(ws), _p(loss), _p(dl)), "uia_dicece_fwd_bwd")
    return loss, dl
