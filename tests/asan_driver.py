"""Executed by tests/test_capi_symbols.py in a subprocess under LD_PRELOAD of the AddressSanitizer runtime: drives the argument-validation
and error paths of the HOST-sanitized build of the C ABI (nextgen-uia_amd/uia_hip/libuia_hip_asan.so, `make -C nextgen-uia_amd/csrc asan`)
on a machine without a GPU.  Every call must return a negative code with a message; a sanitizer report aborts the process."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nextgen-uia_amd", "uia_hip"))
import importlib.util

spec = importlib.util.spec_from_file_location("uia_lib_table", os.path.join(ROOT, "nextgen-uia_amd", "uia_hip", "_lib.py"))
src = open(spec.origin).read().replace("import torch  # noqa", "# (torch is not imported under the sanitizer)  # noqa")
mod = type(sys)("uia_lib_table")
mod.__file__ = spec.origin
exec(compile(src, spec.origin, "exec"), mod.__dict__)

lib = C.CDLL(sys.argv[1])
for name, (res, args) in mod.PROTOTYPES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args


def err():
    return lib.uia_last_error().decode()


checks = 0


def expect_fail(rc, needle=None):
    global checks
    assert rc != 0, "call unexpectedly succeeded"
    msg = err()
    assert msg and len(msg) < 512, msg
    if needle:
        assert needle in msg, (needle, msg)
    checks += 1


FAKE = 0x10000                      # aligned, non-null, never dereferenced on the host
assert lib.uia_version() >= 100
expect_fail(lib.uia_gemm(None, 1, None, 0), "null descriptor")
d = mod.GemmDesc()
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "empty problem")
d.A, d.W, d.M, d.N, d.K, d.lda, d.ldw, d.alpha = FAKE, FAKE, 300, 128, 48, 48, 48, 1.0
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "K=48")
d.K = d.lda = d.ldw = 64
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "no output")
d.outT, d.ldo = FAKE, 64
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "ldo=64 < N=128")
d.ldo = 128
expect_fail(lib.uia_gemm(None, 7, C.byref(d), 0), "bad dtype")
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 99), "unknown tile config")
d.w_kblocked = 1
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 3), "K-blocked")
d.w_kblocked = 0
d.A = FAKE + 2
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "16-byte aligned")
d.A = FAKE
d.dact = 1
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "dact needs aux_in")
d.dact = 0
d.drop_where = 3
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "drop_where=3")
d.drop_where, d.drop_p = 2, 1.0
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "outside [0, 1)")
d.drop_where, d.drop_p = 1, 0.1                                         # dropout on the A operand is the N = 64 stream kernel's: N = 128 here
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "drop_where = 1")
expect_fail(lib.uia_gemm(None, 0, C.byref(d), 0), "dropout on the A operand needs bf16")
d.drop_where, d.a_drop_out = 0, FAKE
expect_fail(lib.uia_gemm(None, 1, C.byref(d), 0), "a_drop_out without drop_where")
d.a_drop_out, d.drop_p = None, 0.0
expect_fail(lib.uia_gemm(None, 0, C.byref(d), 23), "tile cfg 23 is the bf16")
for cfg in (0, 1, 3, 8, 12, 13, 14, 21, 23):                                # valid descriptor: validation passes, the launch finds no device
    expect_fail(lib.uia_gemm(None, 1, C.byref(d), cfg))
expect_fail(lib.uia_wgrad(None, 1, 100, 48, 64, FAKE, 64, FAKE, 64, 1.0, FAKE, None), "multiples of 64")
expect_fail(lib.uia_wgrad_ex(None, 1, 100, 64, 64, FAKE, 64, FAKE, 64, 1.0, FAKE, 0, 16, 64, None), "ldw=0")
expect_fail(lib.uia_wgrad_ex(None, 1, 100, 64, 64, FAKE, 64, FAKE, 64, 1.0, FAKE, 16, 65, 16, None), "valid extent")
expect_fail(lib.uia_wgrad_ex(None, 1, 100, 64, 64, FAKE, 64, FAKE, 64, 1.0, FAKE, 8, 64, 16, None), "valid extent")
a = mod.AttnDesc()
expect_fail(lib.uia_attn_fwd(None, 1, None), "null descriptor")
expect_fail(lib.uia_attn_fwd(None, 1, C.byref(a)), "empty problem")
a.B, a.H, a.L, a.dh = 2, 2, 400, 64
expect_fail(lib.uia_attn_fwd(None, 1, C.byref(a)), "scale must be positive")          # scale = 0 in a zeroed descriptor
a.scale = -0.125
expect_fail(lib.uia_attn_bwd(None, 1, C.byref(a)), "scale must be positive")
a.scale = 0.125
expect_fail(lib.uia_attn_fwd(None, 1, C.byref(a)), "L=400")
a.L = 64
expect_fail(lib.uia_attn_fwd(None, 1, C.byref(a)), "null tensor")
expect_fail(lib.uia_attn_bwd(None, 1, C.byref(a)), "null tensor")
expect_fail(lib.uia_layernorm_fwd(None, 1, 4, 2048, 2048, FAKE, FAKE, FAKE, 1e-5, FAKE, None), "unsupported shape")
expect_fail(lib.uia_layernorm_bwd(None, 1, 4, 64, 32, FAKE, FAKE, FAKE, 1e-5, None, FAKE, None), "row stride")
expect_fail(lib.uia_embed(None, 8, 40, 64, 100, 16, FAKE, FAKE, FAKE, None, FAKE), "position table")
expect_fail(lib.uia_mona_pre_bwd(None, 1, 8, 64, FAKE, FAKE, None, FAKE, FAKE, FAKE, FAKE, 1e-5, None, None, FAKE, FAKE, FAKE, FAKE, None, 0), "null tensor")
assert lib.uia_mona_pre_bwd_workspace_bytes(50432, 768) == 1024 * 4 * 768 * 4
s = mod.MonaSpatialDesc()
expect_fail(lib.uia_mona_spatial_fwd(None, 1, None), "null descriptor")
expect_fail(lib.uia_mona_spatial_fwd(None, 1, C.byref(s)))
f = mod.MonaFusedDesc()
expect_fail(lib.uia_mona_fused_fwd(None, 1, None), "null descriptor")
expect_fail(lib.uia_mona_fused_fwd(None, 1, C.byref(f)), "use the unfused launches")          # zeroed descriptor: unsupported shape
f.sp.B, f.sp.h, f.sp.w, f.sp.bott, f.D = 2, 14, 14, 64, 1024
expect_fail(lib.uia_mona_fused_fwd(None, 1, C.byref(f)), "use the unfused launches")          # ViT-L width
f.D = 768
expect_fail(lib.uia_mona_fused_fwd(None, 0, C.byref(f)), "use the unfused launches")          # fp32
expect_fail(lib.uia_mona_fused_fwd(None, 1, C.byref(f)), "null tensor")
assert lib.uia_mona_fused_supported(1, 768, 14, 14, 64) == 1 and lib.uia_mona_fused_supported(1, 768, 16, 16, 64) == 0
expect_fail(lib.uia_infonce_fwd_bwd(None, 0, 0, None, None, 1.0, 1.0, None, None, None, None, 0))
expect_fail(lib.uia_adamw_clip_step(None, 0, None, None, None, None, 1e-3, 0.9, 0.95, 1e-8, 0.01, 1.0, 1, 1.0, None))
# round 5: the guarded forms (null buffers, empty problem, negative log index / schedule length)
expect_fail(lib.uia_grad_accum_guarded(None, 0, None, None, None, None, None, None, 0), "bad arguments")
expect_fail(lib.uia_grad_accum_guarded(None, 16, 64, 64, 64, 64, 64, 64, -1), "bad arguments")
expect_fail(lib.uia_adamw_clip_step_guarded(None, 0, None, None, None, None, 1e-3, 0.0, 0, 0.9, 0.95, 1e-8, 0.01, 1.0, 1.0, 1.0, None, None), "bad arguments")
expect_fail(lib.uia_adamw_clip_step_guarded(None, 16, 64, 64, 64, 64, 1e-3, 0.0, -3, 0.9, 0.95, 1e-8, 0.01, 1.0, 1.0, 1.0, 64, 64), "bad arguments")
# communicator: argument errors and use-before-init
buf = C.create_string_buffer(8)
expect_fail(lib.uia_comm_get_unique_id(buf, 8), "buffer too small")
expect_fail(lib.uia_comm_init(3, 2, buf, 8), "bad arguments")
expect_fail(lib.uia_allreduce_sum(None, 0, FAKE, 16), "not initialised")
expect_fail(lib.uia_allgather(None, 0, FAKE, FAKE, 16), "not initialised")
assert lib.uia_comm_world() == 1 and lib.uia_comm_initialised() == 0 and lib.uia_comm_destroy() == 0
print(f"asan driver: {checks} error paths exercised, no sanitizer report")
