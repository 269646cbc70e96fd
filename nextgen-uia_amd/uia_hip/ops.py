"""Tensor-level wrappers over the C ABI: torch tensors in, kernels launched on the current stream.

PyTorch is only plumbing here (device memory, streams); there is no arithmetic in this file and no
CPU path: CPU tensors raise.  `dtype` of an op is taken from its operand tensors
(torch.bfloat16 -> UIA_BF16, torch.float32 -> UIA_F32).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import AttnDesc, GemmDesc, UiaError, check, lib

_ACT = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "relu": 3}


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


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rowmajor(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise UiaError(f"{name} must be a 2-D row-major tensor (got shape {tuple(t.shape)}, stride {t.stride()})")
    return t.stride(0)


def gemm(a, w, *, bias=None, act=None, dact=None, aux_in=None, aux_out=None, resid=None, resid_mod=0, resid_row_off=0,
         resid_t=None, out_group=0, out_t=None, out32=None, alpha=1.0, tile_cfg=0):
    """C = epilogue(alpha * a @ w.T).  a [M,K], w [N,K] share a dtype (bf16 | fp32); see include/uia_hip.h."""
    d = GemmDesc()
    d.lda, d.ldw = _rowmajor(a, "a"), _rowmajor(w, "w")
    if a.dtype != w.dtype or a.shape[1] != w.shape[1]:
        raise UiaError(f"gemm operand mismatch: a {tuple(a.shape)} {a.dtype}, w {tuple(w.shape)} {w.dtype}")
    d.A, d.W = _p(a), _p(w)
    d.M, d.K, d.N = a.shape[0], a.shape[1], w.shape[0]
    d.alpha = alpha
    d.act, d.dact = _ACT[act], _ACT[dact]
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == d.N
        d.bias = _p(bias)
    for name, t, want in (("aux_in", aux_in, a.dtype), ("aux_out", aux_out, a.dtype), ("resid_t", resid_t, a.dtype),
                          ("out_t", out_t, a.dtype), ("resid", resid, torch.float32), ("out32", out32, torch.float32)):
        if t is not None and t.dtype != want:
            raise UiaError(f"gemm {name} must be {want}, got {t.dtype}")
    if aux_in is not None:
        d.aux_in, d.ldaux_in = _p(aux_in), _rowmajor(aux_in, "aux_in")
    if aux_out is not None:
        d.aux_out, d.ldaux_out = _p(aux_out), _rowmajor(aux_out, "aux_out")
    if resid is not None:
        d.resid, d.ldr = _p(resid), _rowmajor(resid, "resid")
    d.resid_mod, d.resid_row_off, d.out_group = resid_mod, resid_row_off, out_group
    if resid_t is not None:
        d.residT, d.ldrT = _p(resid_t), _rowmajor(resid_t, "resid_t")
    if out_t is not None:
        d.outT, d.ldo = _p(out_t), _rowmajor(out_t, "out_t")
    if out32 is not None:
        d.out32, d.ldo32 = _p(out32), _rowmajor(out32, "out32")
    check(lib().uia_gemm(_stream(), _code(a.dtype), C.byref(d), tile_cfg), "uia_gemm")


def wgrad(a, b, dw, dbias=None, alpha=1.0):
    """dw[I,J] += alpha * a.T @ b   (a [M,I], b [M,J]); dbias[I] += a.sum(0).  dw/dbias fp32, pre-zeroed or accumulating."""
    lda, ldb = _rowmajor(a, "a"), _rowmajor(b, "b")
    assert a.dtype == b.dtype and a.shape[0] == b.shape[0] and dw.dtype == torch.float32 and dw.is_contiguous()
    assert tuple(dw.shape) == (a.shape[1], b.shape[1])
    check(lib().uia_wgrad(_stream(), _code(a.dtype), a.shape[0], a.shape[1], b.shape[1], _p(a), lda, _p(b), ldb, alpha, _p(dw), _p(dbias)), "uia_wgrad")


def _attn_desc(q, k, v, out, lse, B, H, L, mask, keylen, scale):
    d = AttnDesc()
    for t in (q, k, v):
        if t.stride(-1) != 1 or t.dtype != q.dtype:
            raise UiaError("attention operands must be unit-stride in the last dim and share a dtype")
    d.q, d.k, d.v, d.ld_qkv = _p(q), _p(k), _p(v), q.stride(-2)
    assert k.stride(-2) == q.stride(-2) == v.stride(-2)
    d.out, d.ldo = _p(out), out.stride(-2)
    d.lse = _p(lse)
    d.keylen = _p(keylen)
    d.B, d.H, d.L, d.dh = B, H, L, 64
    d.mask_kind = {"none": 0, None: 0, "causal": 1, "keypad": 2}[mask]
    d.scale = scale if scale is not None else 64 ** -0.5
    return d


def attn_fwd(q, k, v, out, B, H, L, lse=None, mask=None, keylen=None, scale=None):
    """q,k,v: views whose element (b,l,h,d) is at base[(b*L+l)*ld + h*64 + d] (e.g. slices of the fused qkv)."""
    d = _attn_desc(q, k, v, out, lse, B, H, L, mask, keylen, scale)
    check(lib().uia_attn_fwd(_stream(), _code(q.dtype), C.byref(d)), "uia_attn_fwd")


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, L, mask=None, keylen=None, scale=None):
    d = _attn_desc(q, k, v, out, lse, B, H, L, mask, keylen, scale)
    d.dout, d.lddo = _p(dout), dout.stride(-2)
    d.dq, d.dk, d.dv, d.ld_dqkv = _p(dq), _p(dk), _p(dv), dq.stride(-2)
    assert dk.stride(-2) == dq.stride(-2) == dv.stride(-2)
    check(lib().uia_attn_bwd(_stream(), _code(q.dtype), C.byref(d)), "uia_attn_bwd")


def layernorm_fwd(x, gamma, beta, eps, y_t=None, y32=None, rows=None, ldx=None):
    """x fp32 [rows, D] (row stride ldx); y_t (bf16|fp32) and/or y32 compact [rows, D]."""
    D = gamma.numel()
    rows = x.numel() // D if rows is None else rows
    ldx = D if ldx is None else ldx
    dt = _code(y_t.dtype) if y_t is not None else _lib.F32
    check(lib().uia_layernorm_fwd(_stream(), dt, rows, D, ldx, _p(x), _p(gamma), _p(beta), eps, _p(y_t), _p(y32)), "uia_layernorm_fwd")


def layernorm_bwd(dy, x, gamma, eps, dres=None, dx32=None, dx_t=None, rows=None, ldx=None):
    D = gamma.numel()
    rows = dy.numel() // D if rows is None else rows
    ldx = D if ldx is None else ldx
    check(lib().uia_layernorm_bwd(_stream(), _code(dy.dtype), rows, D, ldx, _p(dy), _p(x), _p(gamma), eps, _p(dres), _p(dx32), _p(dx_t)), "uia_layernorm_bwd")


def cast(src, dst, scale=1.0):
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    check(lib().uia_cast(_stream(), _code(dst.dtype), src.numel(), _p(src), _p(dst), scale), "uia_cast")


def transpose_cast(src, dst):
    assert src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous() and dst.is_contiguous()
    assert tuple(dst.shape) == (src.shape[1], src.shape[0])
    check(lib().uia_transpose_cast(_stream(), _code(dst.dtype), src.shape[0], src.shape[1], _p(src), _p(dst)), "uia_transpose_cast")


def im2col(img, out, patch):
    B, Cc, H, W = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous() and out.is_contiguous()
    check(lib().uia_im2col(_stream(), _code(out.dtype), B, Cc, H, W, patch, _p(img), _p(out)), "uia_im2col")


def fill_cls(x, cls, pos0):
    B, N, D = x.shape
    check(lib().uia_fill_cls(_stream(), B, N, D, _p(cls), _p(pos0), _p(x)), "uia_fill_cls")


def embed(ids, table, pos, type0, out):
    rows, L = ids.numel(), ids.shape[-1]
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    check(lib().uia_embed(_stream(), rows, L, table.shape[1], _p(ids), _p(table), _p(pos), _p(type0), _p(out)), "uia_embed")


def gather_rows(src, idx, dst):
    assert idx.dtype == torch.int64
    check(lib().uia_gather_rows(_stream(), idx.numel(), src.shape[-1], _p(src), _p(idx), _p(dst)), "uia_gather_rows")
