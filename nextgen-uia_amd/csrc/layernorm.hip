// layernorm.hip — row LayerNorm over the fp32 residual stream, forward and (frozen-affine) backward.
//
// Replaces: the LayerNorm of /root/reference/src/third_party/openai_clip/model.py:163-169 (eps 1e-5:
//           ln_1 / ln_2 / ln_pre / ln_post / ln_final), timm Block.norm1 / norm2 / trunk.norm
//           (eps 1e-6), HF BERT LayerNorm (eps 1e-12), CLIPSeg decoder LayerNorms.
// The Mona adapter's own (trainable) LayerNorm is fused elsewhere (mona.hip).
//
// HBM-bound: one wave per row, 16-byte loads, two-pass statistics held in registers; the backward
// recomputes mean / rstd from x instead of storing them.  D ≤ 1024, D % 4 == 0.
// x (and, in the backward, dres / dx) may be row-strided (ldx ≥ D): the final LayerNorm runs on the
// CLS rows of the [B, N, D] stream in place; y / dy are compact [M, D].
//   fwd:  y = (x - mean) * rstd * gamma + beta            → T copy (GEMM operand) and/or fp32 copy
//   bwd:  dx = dres + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma
//         → fp32 (residual-gradient stream) and optional T copy (operand of the next dgrad GEMM)
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

constexpr int LN_MAXV = 4;   // float4 per lane → D ≤ 64*4*4 = 1024

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int M, int D, long ldx, const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, T* __restrict__ yT, float* __restrict__ y32,
                                                      float* __restrict__ stats) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = D >> 2;
    const float* xr = x + (size_t)row * ldx;
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < nv ? load4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (stats && lane == 0) *(float2*)(stats + 2 * (size_t)row) = float2{mean, rstd};   // for uia_gemm's deferred LayerNorm residual
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
        if (c < nv) {
            const f32x4 g = load4(gamma + 4 * c), b = load4(beta + 4 * c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf((v[k][e] - mean) * rstd, g[e], b[e]);
            if (yT) store4(yT + (size_t)row * D + 4 * c, y);
            if (y32) store4(y32 + (size_t)row * D + 4 * c, y);
        }
    }
}

// Three-byte tensors (round 4; include/uia_hip.h, uia_gemm_desc.resid_lo8): a value is its bf16 hi plane + a signed low byte,
// float bits = (hi_bits << 16) + (lo << 8).  Inside a frozen block the attention-half output x1 and its gradient dx1 travel in that form
// between the GEMM epilogues and this kernel: 3 bytes read instead of 4, 3 written instead of 4 + 2.
__device__ __forceinline__ f32x4 three_byte_load4(const bf16_t* hi, const int8_t* lo) {
    typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
    const u16x4 h = *(const u16x4*)hi;
    const int l = *(const int*)lo;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(float, ((unsigned)h[e] << 16) + ((unsigned)__builtin_amdgcn_sbfe(l, 8 * e, 8) << 8));
    return r;
}
__device__ __forceinline__ void three_byte_store4(bf16_t* hi, int8_t* lo, f32x4 v) {
    unsigned w[4];
    unsigned short hs[4];
    const float f[4] = {v[0], v[1], v[2], v[3]};      // (bit-casting v[e] of the ext_vector inside the unrolled loop read element 0 for every e: hipcc 7.2)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned vb = __builtin_bit_cast(unsigned, f[e]);
        hs[e] = __builtin_bit_cast(unsigned short, (bf16_t)f[e]);
        int d = (int)((vb + 0x80u) >> 8) - (int)((unsigned)hs[e] << 8);
        d = d < -127 ? -127 : (d > 127 ? 127 : d);
        w[e] = (unsigned)d & 0xFFu;
    }
    uint2 hp;
    hp.x = (unsigned)hs[0] | ((unsigned)hs[1] << 16);
    hp.y = (unsigned)hs[2] | ((unsigned)hs[3] << 16);
    *(uint2*)hi = hp;
    *(unsigned*)lo = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
}

// x: fp32 rows (x) or, x_lo != null, a three-byte tensor (x_hi row-major with D columns or K-blocked with x_kb_rows rows per 32-column block, x_lo
// row-major [M, D]).  dres likewise (dres_hi row-major).  Output: dx32 and / or dxT (bf16 / fp32 T copy), or, dx_lo != null, dxT + dx_lo as a
// three-byte tensor.
// THREE: the three-byte forms are a compile-time variant — as run-time branches they cost the plain fp32 launch 17 % (98.7 -> 116-118 us per image-tower launch).
template <typename T, bool THREE = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int M, int D, long ldx, const T* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, float eps, const float* __restrict__ dres,
                                                      float* __restrict__ dx32, T* __restrict__ dxT,
                                                      const bf16_t* __restrict__ x_hi, const int8_t* __restrict__ x_lo, long x_kb_rows,
                                                      const bf16_t* __restrict__ dres_hi, const int8_t* __restrict__ dres_lo, int8_t* __restrict__ dx_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = D >> 2;
    const float* xr = x + (size_t)row * ldx;
    const T* dyr = dy + (size_t)row * D;
    f32x4 v[LN_MAXV], g[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            if (THREE && x_lo) {
                const bf16_t* hp = x_kb_rows ? x_hi + ((size_t)((4 * c) >> 5) * (size_t)x_kb_rows + row) * 32 + ((4 * c) & 31) : x_hi + (size_t)row * D + 4 * c;
                v[k] = three_byte_load4(hp, x_lo + (size_t)row * D + 4 * c);
            } else {
                v[k] = load4(xr + 4 * c);
            }
        }
        g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            const f32x4 d = load4(dyr + 4 * c), w = load4(gamma + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[k][e] = d[e] * w[e];
        }
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = c < nv ? v[k][e] - mean : 0.f;
            v[k][e] = d;
            q = fmaf(d, d, q);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[k][e] *= rstd;   // xhat
            sg += g[k][e];
            sgx = fmaf(g[k][e], v[k][e], sgx);
        }
    const float mg = wave_sum(sg) / D, mgx = wave_sum(sgx) / D;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
        if (c < nv) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            if (THREE && dres_lo) o = three_byte_load4(dres_hi + (size_t)row * D + 4 * c, dres_lo + (size_t)row * D + 4 * c);
            else if (dres) o = load4(dres + (size_t)row * ldx + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += rstd * (g[k][e] - mg - v[k][e] * mgx);
            if (dx32) store4(dx32 + (size_t)row * ldx + 4 * c, o);
            if constexpr (sizeof(T) == 2) {
                if (THREE && dx_lo) three_byte_store4((bf16_t*)dxT + (size_t)row * D + 4 * c, dx_lo + (size_t)row * D + 4 * c, o);
                else if (dxT) store4(dxT + (size_t)row * ldx + 4 * c, o);
            } else {
                if (dxT) store4(dxT + (size_t)row * ldx + 4 * c, o);
            }
        }
    }
}

}  // namespace

int uia_layernorm_fwd_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const float* x, const float* gamma, const float* beta,
                             float eps, void* yT, float* y32, float* stats) {
    UIA_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "uia_layernorm_fwd: unsupported shape M=%d D=%d", M, D);
    UIA_CHECK_ARG(x && gamma && beta && (yT || y32 || stats), "uia_layernorm_fwd: null tensor");
    UIA_CHECK_ARG(ldx >= D && ldx % 4 == 0, "uia_layernorm_fwd: row stride %ld", ldx);
    UIA_CHECK_ARG(((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)yT | (uintptr_t)y32 | (uintptr_t)stats) % 8 == 0, "uia_layernorm_fwd: alignment");
    const dim3 grid((M + 3) / 4), block(256);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, block, 0, stream, M, D, ldx, x, gamma, beta, eps, (bf16_t*)yT, y32, stats);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, stream, M, D, ldx, x, gamma, beta, eps, (float*)yT, y32, stats);
    else { uia_set_error("uia_layernorm_fwd: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_layernorm_bwd_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const void* dy, const float* x, const float* gamma, float eps,
                             const float* dres, float* dx32, void* dxT) {
    return uia_layernorm_bwd3_launch(stream, dtype, M, D, ldx, dy, x, nullptr, nullptr, 0, gamma, eps, dres, nullptr, nullptr, dx32, dxT, nullptr);
}

int uia_layernorm_bwd3_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const void* dy, const float* x, const void* x_hi, const int8_t* x_lo,
                              long x_kb_rows, const float* gamma, float eps, const float* dres, const void* dres_hi, const int8_t* dres_lo, float* dx32,
                              void* dxT, int8_t* dx_lo) {
    UIA_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "uia_layernorm_bwd: unsupported shape M=%d D=%d", M, D);
    UIA_CHECK_ARG(dy && (x || (x_hi && x_lo)) && gamma && (dx32 || dxT), "uia_layernorm_bwd: null tensor");
    UIA_CHECK_ARG(ldx >= D && ldx % 4 == 0, "uia_layernorm_bwd: row stride %ld", ldx);
    const bool three = x_lo || dres_lo || dx_lo;
    UIA_CHECK_ARG(!three || (dtype == UIA_BF16 && ldx == D), "uia_layernorm_bwd: three-byte tensors need bf16 and compact rows (ldx == D)");
    UIA_CHECK_ARG(!x_lo || (x_hi && !x && (x_kb_rows == 0 || (x_kb_rows >= M && D % 32 == 0)) && ((uintptr_t)x_hi % 8) == 0 && ((uintptr_t)x_lo % 4) == 0),
                  "uia_layernorm_bwd: x as a three-byte tensor needs x_hi + x_lo (and no fp32 x), a K-blocked hi plane of at least M rows and D a multiple of 32");
    UIA_CHECK_ARG(!dres_lo || (dres_hi && !dres && ((uintptr_t)dres_hi % 8) == 0 && ((uintptr_t)dres_lo % 4) == 0), "uia_layernorm_bwd: dres as a three-byte tensor needs dres_hi + dres_lo (and no fp32 dres)");
    UIA_CHECK_ARG(!dx_lo || (dxT && ((uintptr_t)dxT % 8) == 0 && ((uintptr_t)dx_lo % 4) == 0), "uia_layernorm_bwd: dx_lo needs dxT as the hi plane");
    UIA_CHECK_ARG(x_lo || x_kb_rows == 0, "uia_layernorm_bwd: x_kb_rows without a three-byte x");
    const dim3 grid((M + 3) / 4), block(256);
    if (dtype == UIA_BF16 && three) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, true>), grid, block, 0, stream, M, D, ldx, (const bf16_t*)dy, x, gamma, eps, dres, dx32, (bf16_t*)dxT,
                                                       (const bf16_t*)x_hi, x_lo, x_kb_rows, (const bf16_t*)dres_hi, dres_lo, dx_lo);
    else if (dtype == UIA_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, false>), grid, block, 0, stream, M, D, ldx, (const bf16_t*)dy, x, gamma, eps, dres, dx32, (bf16_t*)dxT,
                                                   (const bf16_t*)nullptr, (const int8_t*)nullptr, 0L, (const bf16_t*)nullptr, (const int8_t*)nullptr, (int8_t*)nullptr);
    else if (dtype == UIA_F32) hipLaunchKernelGGL((ln_bwd_kernel<float, false>), grid, block, 0, stream, M, D, ldx, (const float*)dy, x, gamma, eps, dres, dx32, (float*)dxT,
                                                  (const bf16_t*)nullptr, (const int8_t*)nullptr, 0L, (const bf16_t*)nullptr, (const int8_t*)nullptr, (int8_t*)nullptr);
    else { uia_set_error("uia_layernorm_bwd: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}
