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

// x: fp32 rows (x) or, x_lo != null, a three-byte tensor (x_hi row-major with D columns or K-blocked with x_kb_rows rows per 32-column block, x_lo
// row-major [M, D]).  dres likewise (dres_hi row-major or K-blocked with dres_kb_rows rows per block).  Output: dx32 and / or dxT (bf16 / fp32 T copy), or, dx_lo != null, dxT + dx_lo as a
// three-byte tensor.
// THREE: the three-byte forms are a compile-time variant — as run-time branches they cost the plain fp32 launch 17 % (98.7 -> 116-118 us per image-tower launch).
// NV: float4 per lane (3: D <= 768, 4: D <= 1024) — the row lives in registers, and the 768-wide launches of the step should not carry a fourth of them for nothing.
// THREE: every operand of the row, the three-byte residual gradient included (raw planes, three registers per float4, decoded at the end), is REQUESTED in the
// first loop: the kernel lives on bytes in flight per wave, and a residual gradient fetched behind the three wave reductions was a second, exposed round trip
// (126 -> 86 us per image-tower launch at 12 bytes per element; the fp32 form, 16 bytes per element, keeps its late request: 109 us).
template <typename T, bool THREE, int NV>
__global__ __launch_bounds__(256, NV == 4 ? (THREE ? 3 : 5) : 1) void ln_bwd_kernel(int M, int D, long ldx, const T* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, float eps, const float* __restrict__ dres,
                                                      float* __restrict__ dx32, T* __restrict__ dxT,
                                                      const bf16_t* __restrict__ x_hi, const int8_t* __restrict__ x_lo, long x_kb_rows,
                                                      const bf16_t* __restrict__ dres_hi, const int8_t* __restrict__ dres_lo, long dres_kb_rows, int8_t* __restrict__ dx_lo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = D >> 2;
    const float* xr = x + (size_t)row * ldx;
    const T* dyr = dy + (size_t)row * D;
    f32x4 v[NV], g[NV];
    uint2 rhi[THREE ? NV : 1];
    unsigned rlo[THREE ? NV : 1];
    const bool r3 = THREE && dres_lo != nullptr;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (THREE) { rhi[k] = uint2{0u, 0u}; rlo[k] = 0u; }
        if (c < nv) {
            if (THREE && x_lo) {
                const bf16_t* hp = x_kb_rows ? x_hi + ((size_t)((4 * c) >> 5) * (size_t)x_kb_rows + row) * 32 + ((4 * c) & 31) : x_hi + (size_t)row * D + 4 * c;
                v[k] = three_byte_load4(hp, x_lo + (size_t)row * D + 4 * c);
            } else {
                v[k] = load4(xr + 4 * c);
            }
            const f32x4 d = load4(dyr + 4 * c), w = load4(gamma + 4 * c);
            if constexpr (THREE) {
                if (r3) {
                    rhi[k] = *(const uint2*)(dres_kb_rows ? dres_hi + ((size_t)((4 * c) >> 5) * (size_t)dres_kb_rows + row) * 32 + ((4 * c) & 31) : dres_hi + (size_t)row * D + 4 * c);
                    rlo[k] = *(const unsigned*)(dres_lo + (size_t)row * D + 4 * c);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) g[k][e] = d[e] * w[e];
        }
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
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
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[k][e] *= rstd;   // xhat
            sg += g[k][e];
            sgx = fmaf(g[k][e], v[k][e], sgx);
        }
    const float mg = wave_sum(sg) / D, mgx = wave_sum(sgx) / D;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < nv) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (THREE) {
                if (r3) o = three_byte_decode4(rhi[k], rlo[k]);
                else if (dres) o = load4(dres + (size_t)row * ldx + 4 * c);      // fp32 residual gradient beside a three-byte operand or result: the late load of the first form
            } else {
                if (dres) o = load4(dres + (size_t)row * ldx + 4 * c);          // fp32 form: requested here, behind the reductions (requested up front it was 3 % slower: 112 vs 109 us)
            }
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
    return uia_layernorm_bwd3_launch(stream, dtype, M, D, ldx, dy, x, nullptr, nullptr, 0, gamma, eps, dres, nullptr, nullptr, 0, dx32, dxT, nullptr);
}

int uia_layernorm_bwd3_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const void* dy, const float* x, const void* x_hi, const int8_t* x_lo,
                              long x_kb_rows, const float* gamma, float eps, const float* dres, const void* dres_hi, const int8_t* dres_lo, long dres_kb_rows,
                              float* dx32, void* dxT, int8_t* dx_lo) {
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
    UIA_CHECK_ARG(dres_kb_rows == 0 || (dres_lo && dres_kb_rows >= M && D % 32 == 0), "uia_layernorm_bwd: dres_kb_rows needs a three-byte dres, at least M rows per column block and D a multiple of 32");
    const dim3 grid((M + 3) / 4), block(256);
#define UIA_LN_BWD(TT, TH, NVV)                                                                                                              \
    hipLaunchKernelGGL((ln_bwd_kernel<TT, TH, NVV>), grid, block, 0, stream, M, D, ldx, (const TT*)dy, x, gamma, eps, dres, dx32, (TT*)dxT,          \
                       (const bf16_t*)x_hi, x_lo, x_kb_rows, (const bf16_t*)dres_hi, dres_lo, dres_kb_rows, dx_lo)
    const bool narrow = D <= 768;
    if (dtype == UIA_BF16 && three) { if (narrow) UIA_LN_BWD(bf16_t, true, 3); else UIA_LN_BWD(bf16_t, true, 4); }
    else if (dtype == UIA_BF16) { if (narrow) UIA_LN_BWD(bf16_t, false, 3); else UIA_LN_BWD(bf16_t, false, 4); }
    else if (dtype == UIA_F32) { if (narrow) UIA_LN_BWD(float, false, 3); else UIA_LN_BWD(float, false, 4); }
#undef UIA_LN_BWD
    else { uia_set_error("uia_layernorm_bwd: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}
