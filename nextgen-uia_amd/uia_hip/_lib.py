"""ctypes binding of libuia_hip.so (C ABI in include/uia_hip.h).

The product path has no CPU fallback: loading fails loudly if the library has not been built
(`make -C nextgen-uia_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`), and every
wrapper raises if handed a non-CUDA tensor.
"""
import ctypes as C
import os
import re

import torch  # noqa: F401  -- must come first: libuia_hip.so has to bind to the SAME libamdhip64 that PyTorch loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UIA_HIP_LIB") or os.path.join(_HERE, "libuia_hip.so")     # UIA_HIP_LIB: diagnostic builds (tools/abwd_variants.sh)

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICKGELU, ACT_RELU = 0, 1, 2, 3
MASK_NONE, MASK_CAUSAL, MASK_KEYPAD = 0, 1, 2
MONA_VARIANTS = {"baseline": 0, "noise_aware": 1, "freq_enhanced": 2, "hybrid": 3}

vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class PackDesc(C.Structure):
    """uia_pack_desc (include/uia_hip.h): one small trainable matrix and the operand forms to derive from it."""
    _fields_ = [("src", vp), ("row", vp), ("row_kb", vp), ("tr", vp), ("tr_kb", vp), ("rows", i32), ("cols", i32),
                ("rows_pad", i32), ("cols_pad", i32), ("scale", f32), ("reserved_", i32)]


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("W", vp), ("ldw", i64), ("M", i32), ("N", i32), ("K", i32), ("alpha", f32),
                ("bias", vp), ("act", i32), ("dact", i32), ("aux_in", vp), ("ldaux_in", i64), ("aux_out", vp), ("ldaux_out", i64),
                ("resid", vp), ("ldr", i64), ("resid_mod", i32), ("resid_row_off", i32), ("residT", vp), ("ldrT", i64),
                ("out_group", i32), ("outT", vp), ("ldo", i64), ("out32", vp), ("ldo32", i64), ("w_kblocked", i32),
                ("resid_ln_stats", vp), ("resid_ln_w", vp), ("resid_ln_b", vp),
                ("rowsum_out", vp), ("lnfold_sums", vp), ("lnfold_colsum", vp), ("lnfold_dim", i32), ("lnfold_eps", f32),
                ("resid_ln_dim", i32), ("resid_ln_eps", f32), ("a_kb_rows", i64), ("outT_kb_rows", i64), ("ln_flag", vp), ("ln_flag_limit", f32),
                ("drop_where", i32), ("drop_p", f32), ("drop_seed", C.c_uint64), ("a_drop_out", vp), ("splitk_ws", vp),
                ("A2", vp), ("lda2", i64), ("K2", i32), ("a2_group_cols", i32), ("a2_group_stride", i64),
                ("resid_lo8", vp), ("ld_resid_lo", i64), ("residT_kb_rows", i64), ("out_lo8", vp), ("ld_out_lo", i64),
                ("resid_lo_kb_rows", i64), ("out_lo_kb_rows", i64)]


class AttnDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("ld_qkv", i64), ("out", vp), ("ldo", i64), ("lse", vp), ("keylen", vp),
                ("B", i32), ("H", i32), ("L", i32), ("dh", i32), ("mask_kind", i32), ("scale", f32),
                ("dout", vp), ("lddo", i64), ("dq", vp), ("dk", vp), ("dv", vp), ("ld_dqkv", i64), ("cu_seqlens", vp),
                ("out_kb_rows", i64), ("dqkv_kb_rows", i64)]


class MonaSpatialDesc(C.Structure):
    _fields_ = [("variant", i32), ("B", i32), ("h", i32), ("w", i32), ("bott", i32),
                ("t", vp), ("d", vp),
                ("conv1_w", vp), ("conv1_b", vp), ("conv2_w", vp), ("conv2_b", vp), ("conv3_w", vp), ("conv3_b", vp),
                ("proj_w", vp), ("proj_b", vp), ("freq", vp), ("ne1_w", vp), ("ne1_b", vp), ("ne3_w", vp), ("ne3_b", vp),
                ("p_drop", f32), ("seed", C.c_uint64), ("keep_mask", vp),
                ("dd", vp), ("dt", vp),
                ("g_conv1_w", vp), ("g_conv1_b", vp), ("g_conv2_w", vp), ("g_conv2_b", vp), ("g_conv3_w", vp), ("g_conv3_b", vp),
                ("g_proj_w", vp), ("g_proj_b", vp), ("g_freq", vp), ("g_ne1_w", vp), ("g_ne1_b", vp), ("g_ne3_w", vp), ("g_ne3_b", vp), ("ws", vp)]


class WgradGroupDesc(C.Structure):
    """include/uia_hip.h: uia_wgrad_group_desc"""
    _fields_ = [("n", C.c_int32), ("M", C.c_int32), ("I", C.c_int32), ("J", C.c_int32), ("A", C.c_void_p * 4), ("B", C.c_void_p * 4), ("dW", C.c_void_p * 4),
                ("dbias_A", C.c_void_p * 4), ("lda", C.c_int64), ("ldb", C.c_int64), ("ldw", C.c_int64), ("i_valid", C.c_int32), ("j_valid", C.c_int32),
                ("alpha", C.c_float), ("drop_p", C.c_float), ("drop_seed", C.c_uint64 * 4), ("drop_ld", C.c_int64), ("drop_col0", C.c_int32), ("reserved", C.c_int32)]


class LoraRankDesc(C.Structure):
    """uia_lora_rank_desc (include/uia_hip.h): out += sum_i drop_i(alpha * Q_i @ W_i.T), up to three rank-64 sources in one pass."""
    _fields_ = [("M", i32), ("N", i32), ("nsrc", i32), ("alpha", f32), ("Q", vp), ("ldq", i64), ("q_stride", i64), ("W", vp * 3), ("ldw", i64),
                ("out", vp), ("ldo", i64), ("drop_p", f32), ("seed", C.c_uint64 * 3)]


class LnLoraDesc(C.Structure):
    """uia_ln_lora_desc (include/uia_hip.h): LayerNorm + up to three LoRA down-projections of its output in one launch."""
    _fields_ = [("M", i32), ("D", i32), ("nsrc", i32), ("eps", f32), ("x", vp), ("ldx", i64), ("gamma", vp), ("beta", vp), ("h", vp), ("A", vp * 3), ("lda", i64),
                ("T", vp), ("t_stride", i64), ("drop_p", f32), ("seed", C.c_uint64 * 3)]


class MonaFusedDesc(C.Structure):
    """uia_mona_fused_desc (include/uia_hip.h): the whole adapter forward of one image in one workgroup."""
    _fields_ = [("sp", MonaSpatialDesc), ("D", i32), ("eps", f32), ("x", vp), ("norm_w", vp), ("norm_b", vp), ("gamma", vp), ("gammax", vp),
                ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("y32", vp), ("yT", vp), ("yT_kb_rows", i64), ("rowsum_out", vp), ("ln_flag", vp),
                ("u_out", vp), ("t_out", vp)]


# name -> (restype, argtypes).  tests/test_capi_symbols.py checks this table against include/uia_hip.h.
PROTOTYPES = {
    "uia_last_error": (C.c_char_p, []),
    "uia_version": (C.c_int, []),
    "uia_gemm": (C.c_int, [vp, C.c_int, C.POINTER(GemmDesc), C.c_int]),
    "uia_wgrad": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, f32, vp, vp]),
    "uia_wgrad_drop": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, f32, vp, i64, C.c_int, C.c_int, f32, C.c_uint64, i64, C.c_int]),
    "uia_wgrad_group": (C.c_int, [vp, C.c_int, vp]),
    "uia_wgrad_ex": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, f32, vp, i64, C.c_int, C.c_int, vp]),
    "uia_attn_fwd": (C.c_int, [vp, C.c_int, C.POINTER(AttnDesc)]),
    "uia_attn_bwd": (C.c_int, [vp, C.c_int, C.POINTER(AttnDesc)]),
    "uia_attn_bwd_cfg": (C.c_int, [vp, C.c_int, C.POINTER(AttnDesc), C.c_int]),
    "uia_layernorm_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, vp, vp, vp, f32, vp, vp]),
    "uia_layernorm_fwd_stats": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, vp, vp, vp, f32, vp, vp, vp]),
    "uia_layernorm_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, vp, vp, vp, f32, vp, vp, vp]),
    "uia_ln_lora_down": (C.c_int, [vp, C.c_int, C.POINTER(LnLoraDesc)]),
    "uia_lora_rank_update": (C.c_int, [vp, C.c_int, C.POINTER(LoraRankDesc)]),
    "uia_layernorm_bwd3": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, vp, vp, vp, vp, i64, vp, f32, vp, vp, vp, i64, vp, vp, vp]),
    "uia_mona_pre_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, f32, vp]),
    "uia_mona_pre_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, i64]),
    "uia_mona_pre_fwd_t": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, f32, vp, vp, i64, vp, vp, i64]),
    "uia_mona_pre_bwd_du": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, i64]),
    "uia_mona_pre_bwd_du3": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, i64]),
    "uia_mona_pre_bwd_workspace_bytes": (sz, [C.c_int, C.c_int]),
    "uia_mona_spatial_fwd": (C.c_int, [vp, C.c_int, C.POINTER(MonaSpatialDesc)]),
    "uia_mona_spatial_bwd": (C.c_int, [vp, C.c_int, C.POINTER(MonaSpatialDesc)]),
    "uia_mona_spatial_workspace_bytes": (sz, [C.c_int]),
    "uia_mona_fused_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "uia_mona_fused_fwd": (C.c_int, [vp, C.c_int, C.POINTER(MonaFusedDesc)]),
    "uia_upsample_bilinear_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp]),
    "uia_upsample_bilinear_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int64]),
    "uia_segment_mean_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp]),
    "uia_segment_mean_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int64]),
    "uia_dicece_workspace_bytes": (sz, [C.c_int]),
    "uia_dicece_fwd_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp, vp]),
    "uia_infonce_workspace_bytes": (sz, [C.c_int, C.c_int]),
    "uia_infonce_fwd_bwd": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, f32, f32, vp, vp, vp, vp, sz]),
    "uia_adamw_clip_step": (C.c_int, [vp, sz, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, C.c_int, f32, vp]),
    "uia_grad_accum_guarded": (C.c_int, [vp, sz, vp, vp, vp, vp, vp, vp, i64]),
    "uia_adamw_clip_step_guarded": (C.c_int, [vp, sz, vp, vp, vp, vp, f32, f32, C.c_int, f32, f32, f32, f32, f32, f32, f32, vp, vp]),
    "uia_comm_unique_id_bytes": (C.c_int, []),
    "uia_comm_get_unique_id": (C.c_int, [vp, C.c_int]),
    "uia_comm_init": (C.c_int, [C.c_int, C.c_int, vp, C.c_int]),
    "uia_comm_world": (C.c_int, []),
    "uia_comm_initialised": (C.c_int, []),
    "uia_allreduce_sum": (C.c_int, [vp, C.c_int, vp, sz]),
    "uia_allgather": (C.c_int, [vp, C.c_int, vp, vp, sz]),
    "uia_comm_destroy": (C.c_int, []),
    "uia_dropout": (C.c_int, [vp, C.c_int, sz, vp, vp, f32, C.c_uint64, C.c_int]),
    "uia_colsum": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, i64, vp]),
    "uia_layernorm_bwd_affine": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, f32, vp, vp, vp, vp]),
    "uia_film_fwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "uia_film_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "uia_im2col3x3": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "uia_col2im3x3": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "uia_unshuffle": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, f32, vp]),
    "uia_shuffle": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, i64]),
    "uia_act_bwd": (C.c_int, [vp, C.c_int, sz, vp, vp, C.c_int, vp]),
    "uia_cast": (C.c_int, [vp, C.c_int, sz, vp, vp, f32]),
    "uia_transpose_cast": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "uia_pack_weights": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int]),
    "uia_im2col": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "uia_im2col_padded": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int64]),
    "uia_fill_cls": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "uia_embed": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "uia_embed_bwd": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int64]),
    "uia_embed_packed": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "uia_gather_rows": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
}

_lib = None
_BLOCKING = ("uia_comm_init", "uia_comm_destroy", "uia_comm_get_unique_id")


class UiaError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library with prototypes applied."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UiaError(f"{LIB_PATH} not built: run `make -C nextgen-uia_amd/csrc` (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback for the uia hot path.")
        # PyDLL: the interpreter lock is NOT released around a call.  Every entry point but the communicator's set-up is an asynchronous enqueue of a few
        # microseconds; releasing the lock ~600 times per training step hands it to whatever other thread wants it (a loader thread unpickling a batch) and the
        # enqueuing thread then waits for it to come back each time (the fine-tune CLI measured 46 ms of wall time per step for 15 ms of CPU).  The calls that can
        # block on other processes (RCCL bootstrap / teardown) go through a CDLL handle of the same library, which does release it.  UIA_CTYPES_RELEASE_GIL=1: the
        # old behaviour, for A/B runs.
        release = os.environ.get("UIA_CTYPES_RELEASE_GIL", "0") == "1"
        handle = (C.CDLL if release else C.PyDLL)(LIB_PATH)
        blocking = C.CDLL(LIB_PATH)
        _one_copy_of("librccl")                  # the library's RCCL must be the one PyTorch loaded (same SONAME: the loader re-uses it
        _one_copy_of("libamdhip64")              # when torch is imported first); two runtimes in one process would not share state
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(blocking if name in _BLOCKING else handle, name)
            fn.restype, fn.argtypes = res, args
            if name in _BLOCKING:
                setattr(handle, name, fn)
        _lib = handle
    return _lib


def _one_copy_of(stem):
    """Fail loudly if two different files of a runtime library are mapped into this process (e.g. /opt/rocm/lib/librccl.so beside
    torch/lib/librccl.so): collectives or streams created through one copy are invisible to the other."""
    try:
        pat = re.compile(r"^" + re.escape(stem) + r"\.so(\.\d+)*$")         # the runtime itself, not plugins that share its prefix
        paths = {line.split()[-1] for line in open("/proc/self/maps") if pat.match(os.path.basename(line.split()[-1]) if line.split() else "")}
    except OSError:
        return
    real = {os.path.realpath(p) for p in paths}
    if len(real) > 1:
        raise UiaError(f"two copies of {stem} are loaded ({sorted(real)}): import torch before uia_hip so that libuia_hip.so binds to PyTorch's runtime")


def check(rc, what=""):
    if rc != 0:
        raise UiaError(f"{what} failed (rc={rc}): {lib().uia_last_error().decode()}")
