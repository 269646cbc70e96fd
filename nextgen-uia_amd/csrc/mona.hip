// mona.hip — the Mona adapter's non-GEMM stages, all four variants, forward and backward.
//
// Reference: /root/reference/src/adapters/mona.py
//   BaselineMona :96-151 (op :75-93)         NoiseAwareMona :198-253 (op :159-195)
//   FreqEnhancedMona :298-362 (op :261-295)  HybridNoiseFreqMona :427-487 (op :370-424)
// Equations: SURVEY.md Appendix E.1.  The adapter is split around its two skinny GEMMs
// (project1 768→64 and project2 64→768 run on gemm.hip, their weight gradients on wgrad.hip):
//
//   mona_pre      u = LN(x)·γ + x·γx                       [M,D] fp32 → T      (mona.py:125)
//   mona_spatial  t → z → d = drop(gelu(z))                per image, [1+hw, 64] tile in LDS
//                 · rfft2·f_c·irfft2 ≡ per-channel scale f_c (freq_filter multiplies every bin of a channel)
//                 · (DW3+DW5+DW7)/3 — or the per-image softmax-weighted sum of the noise variants —
//                   is evaluated as ONE merged 7×7 depth-wise stencil built per image
//                 · 1×1 projector 64×64 with residual, CLS token bypasses the spatial op, exact-erf GELU,
//                   dropout p=0.1 from a counter-based hash (or an explicit keep mask for parity tests)
//   mona_spatial_bwd recomputes c and z from t (no stash), then back-propagates to dt and to every
//                 adapter_conv parameter (atomic fp32 accumulation, one add per parameter per image)
//   mona_pre_bwd  dX = dY + du·γx + LN'(du·γ·w_n);  dγ, dγx, dw_n, db_n column sums
//
// All of it is HBM/LDS-bound VALU work: thread = (channel = lane, pixel group = wave) so that every
// LDS access of a wave is 64 consecutive floats (conflict-free) and boundary tests are wave-uniform.
#include <stdlib.h>
#include <type_traits>
#include "uia_common.h"
#include "uia_kernels.h"

#include "mona_spatial.h"

namespace {
using namespace uia_mona;

constexpr int LN_MAXV = 4;

// ======================================================================================= pre
template <typename T>
__global__ __launch_bounds__(256) void mona_pre_fwd_kernel(int M, int D, const float* __restrict__ x, const float* __restrict__ nw,
                                                            const float* __restrict__ nb, const float* __restrict__ gamma,
                                                            const float* __restrict__ gammax, float eps, T* __restrict__ u) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = D >> 2;
    const float* xr = x + (size_t)row * D;
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
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = lane + 64 * k;
        if (c < nv) {
            const f32x4 w = load4(nw + 4 * c), b = load4(nb + 4 * c), g = load4(gamma + 4 * c), gx = load4(gammax + 4 * c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float n = fmaf((v[k][e] - mean) * rstd, w[e], b[e]);
                y[e] = fmaf(n, g[e], v[k][e] * gx[e]);
            }
            store4(u + (size_t)row * D + 4 * c, y);
        }
    }
}

// The same with project1 inside (round 4, bf16, bottleneck 64, D = 768): t = u·W1ᵀ + b1 (mona.py:126-127) was the N = 64 stream kernel reading back the u this
// kernel had just written.  Here a block walks 16-row tiles: every wave writes its four u rows to global memory (the backward's weight gradient reads them) AND
// to a bf16 LDS tile; then wave w multiplies the tile with its sixteen rows of W1 — its 24 A fragments stay in registers for the whole launch (96 VGPRs: the
// row pass needs 12) — in the k order of the stream kernel it replaces: t is bit-identical.  LDS rows are padded by 16 bytes: the sixteen rows of a
// ds_read_b128 land on sixteen different bank quads.
template <int KS>                                              // K steps of 32 columns: D = 32·KS
__global__ __launch_bounds__(256) void mona_pre_fwd_t_kernel(int M, const float* __restrict__ x, const float* __restrict__ nw, const float* __restrict__ nb,
                                                              const float* __restrict__ gamma, const float* __restrict__ gammax, float eps,
                                                              bf16_t* __restrict__ u, const bf16_t* __restrict__ w1, long ldw1, const float* __restrict__ b1,
                                                              bf16_t* __restrict__ t, long ldt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 32 * KS, NVF = D / 256, ROWB = 2 * D + 16;
    static_assert(D % 256 == 0, "three float4 per lane and row");
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 wf[KS];                                              // W1 rows 16w + li (the MFMA A operand: rows = columns of t), bytes 64ks + 16g
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *(const uint4*)(w1 + (size_t)(16 * wave + li) * ldw1 + 32 * ks + 8 * g);
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (b1) bias = *(const f32x4*)(b1 + 16 * wave + 4 * g);
    const int ntiles = (M + 15) >> 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = 16 * tile;
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int rl = 4 * wave + i, row = r0 + rl;
            char* urow = smem + rl * ROWB;
            if (row >= M) {                                    // rows past the end: zeros (their t rows are not stored)
#pragma unroll
                for (int k = 0; k < NVF; ++k) *(uint2*)(urow + 8 * (lane + 64 * k)) = uint2{0u, 0u};
                continue;
            }
            const float* xr = x + (size_t)row * D;
            f32x4 v[NVF];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NVF; ++k) { v[k] = load4(xr + 4 * (lane + 64 * k)); s += v[k][0] + v[k][1] + v[k][2] + v[k][3]; }      // the association of mona_pre_fwd_kernel: u is bit-identical
            const float mean = wave_sum(s) / D;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NVF; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q = fmaf(d, d, q); }
            const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
            for (int k = 0; k < NVF; ++k) {
                const int c = lane + 64 * k;
                const f32x4 w = load4(nw + 4 * c), b = load4(nb + 4 * c), gm = load4(gamma + 4 * c), gx = load4(gammax + 4 * c);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = fmaf((v[k][e] - mean) * rstd, w[e], b[e]);
                    y[e] = fmaf(n, gm[e], v[k][e] * gx[e]);
                }
                store4(u + (size_t)row * D + 4 * c, y);
                store4((bf16_t*)(urow + 8 * c), y);
            }
        }
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                      // D[i = t column 16w + 4g + r][j = row li]
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 uf = *(const uint4*)(smem + li * ROWB + 64 * ks + 16 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, uf), acc, 0, 0, 0);
        }
        if (r0 + li < M) {
            const f32x4 o = {acc[0] + bias[0], acc[1] + bias[1], acc[2] + bias[2], acc[3] + bias[3]};
            store4(t + (size_t)(r0 + li) * ldt + 16 * wave + 4 * g, o);
        }
        __syncthreads();
    }
}

// rows are dealt to waves in a grid-stride loop so that every wave keeps per-column partial sums in registers; one LDS reduction
// and one partial row per block.  With u = n·γ + x·γx, n = x̂·w + b  (mona.py:118-124) the four parameter gradients are
//     dγ = w·S1 + b·S0,   dw = γ·S1,   db = γ·S0,   dγx = S2      with  S0 = Σ_m du,  S1 = Σ_m du·x̂,  S2 = Σ_m du·x,
// so the row loop carries THREE column sums and needs two parameter vectors (γ·w and γx); the reduce kernel applies w, b, γ.
// (The first version carried dγ, dγx, dw, db and read w, b, γ, γx per row: 184 VGPRs, two waves per SIMD, 159 us for 619 MB.)
// NV = float4 per lane (D ≤ 256·NV): the common D = 768 runs with NV = 3.
// FUSE (round 4, bf16): du = dt·W1 — project1's data gradient, a K = 64 GEMM that wrote 77 MB for this kernel to read back — is computed HERE: the block
// walks 16-row tiles; its four waves multiply the tile's dt rows [16 x 64] with W1ᵀ ([D, 64] row-major, L2-resident) on the matrix cores, a quarter of the
// columns each, and leave the bf16 tile in LDS in the row layout the row passes read (MFMA with W1ᵀ as the A operand: a lane gets four consecutive columns
// of one row — one 8-byte LDS store per tile); same products in the same order and the same rounding to bf16 as the GEMM launch it replaces.
// (Round 4, measured and removed: the three column sums in LDS with ds_add_f32 instead of 36 registers — 128 VGPRs, four waves per SIMD — took 664 us against 157:
// a 64-lane LDS float atomic retires at ~190 cycles per instruction.  Without the sums at all the three-byte form runs in 128 us, tools/mpb_variants.sh.)
// R3 (round 4, with FUSE): the residual gradient travels as a THREE-BYTE tensor (uia_gemm_desc.resid_lo8) on both sides — dy arrives as (dy_hi row-major, dy_lo),
// dx leaves as (dxT, dx_lo) with dxT, row-major or K-blocked, as its hi plane: 3 bytes read and 3 written per element instead of 4 and 4 + 2.
#ifndef MPB_WAVES
#define MPB_WAVES 3
#endif
template <typename T, int NV, bool KB = false, bool FUSE = false, bool R3 = false>
__global__ __launch_bounds__(256, FUSE ? MPB_WAVES : 1) void mona_pre_bwd_kernel(int M, int D, const T* __restrict__ du, const float* __restrict__ x,
                                                            const float* __restrict__ dy, const float* __restrict__ nw,
                                                            const float* __restrict__ gamma, const float* __restrict__ gammax, float eps,
                                                            float* __restrict__ dx32, T* __restrict__ dxT, float* __restrict__ ws, long dxT_kb,
                                                            const T* __restrict__ dtp, long ldt, const T* __restrict__ w1t, long ldw1,
                                                            const bf16_t* __restrict__ dy_hi, const int8_t* __restrict__ dy_lo, int8_t* __restrict__ dx_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform row → scalar row pointers
    // KB: the T copy of dx is written K-blocked ([D/g][dxT_kb rows][g], g = 64 bytes of elements).  A compile-time variant: the address
    // arithmetic behind a run-time flag cost the row-major kernel 14 registers and with them its third wave per SIMD (105 -> 126 us).
    // Lane l's four columns 4(l + 64k) lie in column block l/8 + 8k (bf16: l/4 + 16k for fp32) at (4l) % g: a per-lane element offset
    // that fits 32 bits, plus a wave-uniform (row, k) part.
    constexpr int KBG = 64 / (int)sizeof(T), LPB = KBG / 4;   // lanes per column block
    const int kb_lane = KB ? (lane / LPB) * (int)dxT_kb * KBG + (lane % LPB) * 4 : 0;
    const int nv = D >> 2;
    // The parameter vectors live in LDS ([2][D] floats after the reduction area) and are re-read per row through an opaque offset:
    // left to itself the compiler hoists them out of the row loop and the kernel drops to two waves per SIMD.
    float* prm = (float*)smem + 6 * D;
    for (int i = threadIdx.x; i < D; i += 256) {
        prm[i] = gamma[i] * nw[i];
        prm[D + i] = gammax[i];
    }
    __syncthreads();
    struct Raw3 { uint2 h; unsigned l; };
    using OT = std::conditional_t<R3, Raw3, f32x4>;             // a row's dy: fp32 values, or the raw planes of a three-byte tensor (three registers per float4)
    const bool want_dx = R3 || dx32 != nullptr || dxT != nullptr;
    f32x4 a0[NV], a1[NV], a2[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) a0[k] = a1[k] = a2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // x, du and dy of a row are requested together, and the NEXT row of the wave is requested before the current one is processed
    // (two register sets, the loop is unrolled by two): at four waves per SIMD one row in flight per wave left HBM at 4.6 TB/s.
    auto issue = [&](int row, f32x4 (&v)[NV], f32x4 (&d)[NV], OT (&o)[NV]) {
        const float* xr = x + (size_t)row * D;
        const T* dur = FUSE ? nullptr : du + (size_t)row * D;
        const float* dyr = R3 ? nullptr : dy + (size_t)row * D;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            const bool ok = c < nv;
            v[k] = ok ? load4(xr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!FUSE) d[k] = ok ? load4(dur + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (R3) {
                // raw planes, decoded in process(): three registers in flight per float4 instead of four
                const uint2 hraw = ok ? *(const uint2*)(dy_hi + (size_t)row * D + 4 * c) : uint2{0u, 0u};
                const unsigned lraw = ok ? *(const unsigned*)(dy_lo + (size_t)row * D + 4 * c) : 0u;
                o[k] = OT{hraw, lraw};
            } else {
                if (want_dx) o[k] = ok ? load4(dyr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto process = [&](int row, f32x4 (&v)[NV], f32x4 (&d)[NV], OT (&o)[NV]) {
        int poff = 0;
        asm volatile("" : "+v"(poff));                            // keeps the parameter reads inside the loop
        const float* pr = prm + poff;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = c < nv ? v[k][e] - mean : 0.f; v[k][e] = t; q = fmaf(t, t, q); }   // v := x − mean
        }
        const float rstd = rsqrtf(wave_sum(q) / D + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            const f32x4 pgw = c < nv ? *(const f32x4*)(pr + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xhat = v[k][e] * rstd;
#ifndef MPB_NO_ACC      // diagnostic builds (tools/mpb_variants.sh): the kernel without its 36 accumulator registers (gradients WRONG)
                a0[k][e] += d[k][e];                                       // S0
                a1[k][e] = fmaf(d[k][e], xhat, a1[k][e]);                  // S1
                a2[k][e] = fmaf(d[k][e], v[k][e] + mean, a2[k][e]);        // S2 (x itself)
#endif
                const float gv = d[k][e] * pgw[e];                         // dL/dx̂
                sg += gv;
                sgx = fmaf(gv, xhat, sgx);
                v[k][e] = xhat;                                            // v := x̂
            }
        }
        const float mg = wave_sum(sg) / D, mgx = wave_sum(sgx) / D;
        if (want_dx) {
            float* dx32r = dx32 ? dx32 + (size_t)row * D : nullptr;
            T* dxTr = dxT ? dxT + (size_t)row * D : nullptr;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                if (c < nv) {
                    const f32x4 pgw = *(const f32x4*)(pr + 4 * c), pgx = *(const f32x4*)(pr + D + 4 * c);
                    f32x4 r;
                    f32x4 oo;
                    if constexpr (R3) oo = three_byte_decode4(o[k].h, o[k].l);      // (o holds the raw planes as integers: bit-casting elements of a float ext_vector reads element 0 for every e, hipcc 7.2)
                    else oo = o[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gv = d[k][e] * pgw[e];
                        r[e] = oo[e] + fmaf(d[k][e], pgx[e], rstd * (gv - mg - v[k][e] * mgx));
                    }
                    if constexpr (R3) {
                        bf16_t* hp = KB ? (bf16_t*)dxT + ((size_t)(k * (64 / LPB)) * (size_t)dxT_kb + row) * KBG + kb_lane : (bf16_t*)dxT + (size_t)row * D + 4 * c;
                        three_byte_store4(hp, dx_lo + (size_t)row * D + 4 * c, r);
                    } else {
                        if (dx32r) store4(dx32r + 4 * c, r);
                        if (KB) store4(dxT + ((size_t)(k * (64 / LPB)) * (size_t)dxT_kb + row) * KBG + kb_lane, r);
                        else if (dxTr) store4(dxTr + 4 * c, r);
                    }
                }
            }
        }
    };
    f32x4 vA[NV], dA[NV], vB[NV], dB[NV];
    OT oA[NV], oB[NV];
    if constexpr (FUSE) {
        char* du_s = smem + (size_t)8 * D * sizeof(float);             // [16][D] bf16: the tile's du rows, behind the reduction area and the parameter vectors
        const int ntiles = (M + 15) >> 4;
        const int li = lane & 15, g = lane >> 4;
        const int cw = D >> 2;                                         // columns per wave in the MFMA phase (a multiple of 16: D % 64 == 0)
        auto from_lds = [&](int rl, f32x4 (&d)[NV]) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = lane + 64 * k;
                d[k] = c < nv ? load4((const T*)(du_s + ((size_t)rl * D + 4 * c) * sizeof(T))) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        int tile = blockIdx.x;
        if (tile < ntiles && 16 * tile + 4 * wave < M) issue(16 * tile + 4 * wave, vA, dA, oA);
        for (; tile < ntiles; tile += gridDim.x) {
            const int r0 = 16 * tile;
            {   // ---- du tile on the matrix cores: D[i = column 4g + r][j = row li] = Σ_k W1ᵀ[col0 + i][k]·dt[r0 + j][k]
                const int rr = r0 + li < M ? r0 + li : M - 1;
                const T* dtr = dtp + (size_t)rr * ldt + 8 * g;
                uint4 b0 = *(const uint4*)dtr, b1 = *(const uint4*)(dtr + 32);
                if (r0 + li >= M) b0 = b1 = uint4{0u, 0u, 0u, 0u};
                const T* wbase = w1t + (size_t)(wave * cw + li) * ldw1 + 8 * g;
                for (int c0 = 0; c0 < cw; c0 += 64) {                  // four column tiles per batch: eight 16-byte loads in flight
                    uint4 a[4][2];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const T* wr = wbase + (size_t)(c0 + 16 * c) * ldw1;
                        a[c][0] = *(const uint4*)wr;
                        a[c][1] = *(const uint4*)(wr + 32);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[c][0]), __builtin_bit_cast(bf16x8, b0), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[c][1]), __builtin_bit_cast(bf16x8, b1), acc, 0, 0, 0);
                        store4((T*)(du_s + ((size_t)li * D + wave * cw + c0 + 16 * c + 4 * g) * sizeof(T)), acc);
                    }
                }
            }
            __syncthreads();
            const int nt = tile + gridDim.x;
            const int rw = r0 + 4 * wave;                              // this wave's four rows of the tile; the first row of its next tile is requested under the last one
            const int rnext = nt < ntiles ? 16 * nt + 4 * wave : M;
#pragma clang loop unroll(disable)
            for (int i = 0; i < 4; i += 2) {                           // rolled: one copy of the two row passes (unrolled four times the kernel took 240 VGPRs)
                const int ra = rw + i, rb = rw + i + 1, rc = i == 0 ? rw + 2 : rnext;
                if (rb < M) issue(rb, vB, dB, oB);
                if (ra < M) { from_lds(4 * wave + i, dA); process(ra, vA, dA, oA); }
                if (rc < M) issue(rc, vA, dA, oA);
                if (rb < M) { from_lds(4 * wave + i + 1, dB); process(rb, vB, dB, oB); }
            }
            __syncthreads();                                           // every wave is done with the tile's du rows
        }
    } else {
    const int rstep = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    if (row < M) issue(row, vA, dA, oA);
    while (row < M) {
        const int rowB = row + rstep;
        if (rowB < M) issue(rowB, vB, dB, oB);
        process(row, vA, dA, oA);
        if (rowB >= M) break;
        row = rowB + rstep;
        if (row < M) issue(row, vA, dA, oA);
        process(rowB, vB, dB, oB);
    }
    }
    // block reduction in two rounds through [2][3 sums][D] (waves 2,3 park, waves 0,1 add on top): 8·D floats of LDS with the parameter
    // vectors, so that four workgroups fit a CU beside the 120 VGPRs
    float* red = (float*)smem;
    float* mine = red + (wave & 1) * 3 * D;
    if (wave >= 2) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            if (c < nv) { store4(mine + 4 * c, a0[k]); store4(mine + D + 4 * c, a1[k]); store4(mine + 2 * D + 4 * c, a2[k]); }
        }
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            if (c < nv) {
                store4(mine + 4 * c, load4(mine + 4 * c) + a0[k]);
                store4(mine + D + 4 * c, load4(mine + D + 4 * c) + a1[k]);
                store4(mine + 2 * D + 4 * c, load4(mine + 2 * D + 4 * c) + a2[k]);
            }
        }
    }
    __syncthreads();
    // One partial row [3][D] per block, summed over the blocks by mona_pre_reduce_kernel.  (The very first version added every block's
    // row straight into the gradient vectors: 1024 blocks x 3072 float atomics onto the SAME 3072 addresses, all issued as the
    // blocks finish together — the one-row contention case of the microarchitecture guide, 14x below the atomic rate.)
    float* wrow = ws + (size_t)blockIdx.x * 3 * D;
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        wrow[i] = red[i] + red[3 * D + i];
    }
}

// Sums the per-block rows [S0 | S1 | S2] and applies the parameter factors: grid (ceil(3D/256), NSPLIT); each block sums a slice of the
// rows (coalesced: consecutive threads = consecutive columns) and adds its slice's contribution with ONE atomic per output element.
constexpr int PRE_RED_SPLIT = 16;
__global__ __launch_bounds__(256) void mona_pre_reduce_kernel(int nblocks, int D, const float* __restrict__ ws, const float* __restrict__ nw,
                                                               const float* __restrict__ nb, const float* __restrict__ gamma,
                                                               float* __restrict__ g_gamma, float* __restrict__ g_gammax,
                                                               float* __restrict__ g_nw, float* __restrict__ g_nb) {
    const int i = blockIdx.x * 256 + threadIdx.x;                  // (sum index qn, column c)
    if (i >= 3 * D) return;
    const int per = (nblocks + PRE_RED_SPLIT - 1) / PRE_RED_SPLIT;
    const int b0 = blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    if (b1 <= b0) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += ws[(size_t)(b + k) * 3 * D + i];
    }
    for (; b < b1; ++b) a[0] += ws[(size_t)b * 3 * D + i];
    const float S = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int qn = i / D, c = i - qn * D;
    if (qn == 0) { atomicAdd(g_gamma + c, nb[c] * S); atomicAdd(g_nb + c, gamma[c] * S); }          // S0 = Σ du
    else if (qn == 1) { atomicAdd(g_gamma + c, nw[c] * S); atomicAdd(g_nw + c, gamma[c] * S); }     // S1 = Σ du·x̂
    else atomicAdd(g_gammax + c, S);                                                                // S2 = Σ du·x
}

// ======================================================================================= spatial
struct NoiseState {   // per-image noise-estimator activations kept for the backward
    float w[3];
};

// c[pix][ch] = f·Σ K·t[nbr] + Σ w_k b_k + t[pix]   for this thread's channel and pixel group
__device__ __forceinline__ void conv_forward(const float* tS, float* cS, const float* k1, const float* k2, const float* k3, float w1,
                                             float w2, float w3, float bm, float f, int h, int w, int p0, int p1, int c) {
    for (int px = p0; px < p1; ++px) {
        const int y = px / w, x = px - y * w;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int yy = y + i - 3;
            const bool vy = yy >= 0 && yy < h;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int xx = x + j - 3;
                const bool ok = vy && xx >= 0 && xx < w;                   // wave-uniform: no branch, out-of-range taps read token 0 and are zeroed
                const float tv = tS[(ok ? 1 + yy * w + xx : 0) * BOTT + c];
                acc = fmaf(KM(i, j), ok ? tv : 0.f, acc);
            }
        }
        cS[px * BOTT + c] = fmaf(f, acc, bm) + tS[(1 + px) * BOTT + c];
    }
}

__device__ __forceinline__ void grad_out(float* wsrow, int ws_off, float* gptr, int idx, float v) {
    if (wsrow) wsrow[ws_off + idx] = v;           // one writer per element: plain store, summed over images by mona_ws_reduce_kernel
    else atomicAdd(gptr + idx, v);
}

template <typename T, bool BWD>
__global__ __launch_bounds__(512) void mona_spatial_kernel(const uia_mona_spatial_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = p.h, w = p.w, hw = h * w, ntok = hw + 1;
    float* tS = (float*)smem;                 // [ntok][64]
    float* cS = tS + ntok * BOTT;             // [hw][64]   (+64 slack)
    const int cs_floats = BWD ? max(ntok * BOTT, RED_FLOATS) : ntok * BOTT;
    float* gS = cS + cs_floats;               // [ntok][64] backward only
    float* scr = BWD ? gS + ntok * BOTT : cS + cs_floats;
    const int tid = threadIdx.x, c = tid & 63, grp = tid >> 6;
    const int b = blockIdx.x;
    const bool has_freq = p.variant == UIA_MONA_FREQ_ENHANCED || p.variant == UIA_MONA_HYBRID;
    const bool has_noise = p.variant == UIA_MONA_NOISE_AWARE || p.variant == UIA_MONA_HYBRID;
    const int ppg = (hw + NGRP - 1) / NGRP, p0 = grp * ppg, p1 = min(hw, p0 + ppg);
    const size_t tok0 = (size_t)b * ntok;
    float* wsrow = (BWD && p.ws) ? p.ws + (size_t)b * WS_ROW : nullptr;

    load_tokens((const T*)p.t + tok0 * BOTT, tS, ntok, tid);
    __syncthreads();
    const float f = has_freq ? p.freq[c] : 1.0f;
    float w1 = 1.f / 3.f, w2 = 1.f / 3.f, w3 = 1.f / 3.f;
    if (has_noise) {
        noise_forward(p, tS, f, hw, ppg, c, grp, tid, scr);
        w1 = scr[SCR_W]; w2 = scr[SCR_W + 1]; w3 = scr[SCR_W + 2];
    }
    float k1[9], k2[25], k3[49];
#pragma unroll
    for (int i = 0; i < 9; ++i) k1[i] = p.conv1_w[c * 9 + i];
#pragma unroll
    for (int i = 0; i < 25; ++i) k2[i] = p.conv2_w[c * 25 + i];
#pragma unroll
    for (int i = 0; i < 49; ++i) k3[i] = p.conv3_w[c * 49 + i];
    const float b1 = p.conv1_b[c], b2 = p.conv2_b[c], b3 = p.conv3_b[c];
    const float bm = w1 * b1 + w2 * b2 + w3 * b3;
    conv_forward(tS, cS, k1, k2, k3, w1, w2, w3, bm, f, h, w, p0, p1, c);
    __syncthreads();

    // ---- projector + GELU (+dropout): thread = (out channel c, pixel group)
    const float inv_keep = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t thresh = p.p_drop > 0.f ? (uint32_t)fminf(p.p_drop * 4294967296.0f, 4294967295.0f) : 0u;
    auto keep_scale = [&](int tok) -> float {
        const size_t idx = (tok0 + tok) * BOTT + c;
        if (p.keep_mask) return p.keep_mask[idx] ? inv_keep : 0.f;
        if (p.p_drop > 0.f) return dropout_keep(p.seed, (uint32_t)idx, thresh) ? inv_keep : 0.f;
        return 1.0f;
    };
    {
        float prow[64];
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
            const f32x4 v = load4(p.proj_w + c * 64 + k);
            prow[k] = v[0]; prow[k + 1] = v[1]; prow[k + 2] = v[2]; prow[k + 3] = v[3];
        }
        const float pb = p.proj_b[c];
        for (int px = p0; px < p1; ++px) {
            float s = pb;
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                const f32x4 v = *(const f32x4*)(cS + px * BOTT + k);     // broadcast read
                s = fmaf(prow[k], v[0], s); s = fmaf(prow[k + 1], v[1], s); s = fmaf(prow[k + 2], v[2], s); s = fmaf(prow[k + 3], v[3], s);
            }
            const float z = cS[px * BOTT + c] + s;
            const float ks = keep_scale(1 + px);
            if (!BWD) {
                ((T*)p.d)[(tok0 + 1 + px) * BOTT + c] = from_f32<T>(gelu_erf(z) * ks);
            } else {
                const float dd = to_f32(((const T*)p.dd)[(tok0 + 1 + px) * BOTT + c]);
                gS[(1 + px) * BOTT + c] = dd * ks * dgelu_erf(z);         // dz = dp
            }
        }
        if (grp == 0) {   // CLS token bypasses the spatial op (mona.py:132,139)
            const float z = tS[c], ks = keep_scale(0);
            if (!BWD) ((T*)p.d)[tok0 * BOTT + c] = from_f32<T>(gelu_erf(z) * ks);
            else gS[c] = to_f32(((const T*)p.dd)[tok0 * BOTT + c]) * ks * dgelu_erf(z);
        }
    }
    if (!BWD) return;

    // ======================================================================== backward
    __syncthreads();
    // ---- dP[co][ci] += Σ_pix dp[pix][co]·c[pix][ci];  db_p[co] += Σ_pix dp[pix][co]
    {
        const int co = tid >> 3, cb = (tid & 7) * 8;
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = 0.f;
        float sb = 0.f;
        for (int px = 0; px < hw; ++px) {
            const float dpv = gS[(1 + px) * BOTT + co];
            sb += dpv;
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                const f32x4 v = *(const f32x4*)(cS + px * BOTT + cb + k);
                a[k] = fmaf(dpv, v[0], a[k]); a[k + 1] = fmaf(dpv, v[1], a[k + 1]); a[k + 2] = fmaf(dpv, v[2], a[k + 2]); a[k + 3] = fmaf(dpv, v[3], a[k + 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) grad_out(wsrow, WS_PROJ_W, p.g_proj_w, co * 64 + cb + k, a[k]);
        // db_p: the 8 threads of a `co` hold the same sum
        if ((tid & 7) == 0) grad_out(wsrow, WS_PROJ_B, p.g_proj_b, co, sb);
    }
    __syncthreads();
    // ---- dc = dp + Pᵀ·dp, in place (rows of a pixel group belong to one wave)
    {
        float pcol[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) pcol[k] = p.proj_w[k * 64 + c];
        for (int px = p0; px < p1; ++px) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                const f32x4 v = *(const f32x4*)(gS + (1 + px) * BOTT + k);
                s = fmaf(pcol[k], v[0], s); s = fmaf(pcol[k + 1], v[1], s); s = fmaf(pcol[k + 2], v[2], s); s = fmaf(pcol[k + 3], v[3], s);
            }
            const float dcv = gS[(1 + px) * BOTT + c] + s;
            __builtin_amdgcn_wave_barrier();
            gS[(1 + px) * BOTT + c] = dcv;
        }
    }
    __syncthreads();
    // ---- pass A: stencil weight gradients (and the mixing-weight gradients of the noise variants)
    float dkm[49];
#pragma unroll
    for (int i = 0; i < 49; ++i) dkm[i] = 0.f;
    float sdc = 0.f, dwm1 = 0.f, dwm2 = 0.f, dwm3 = 0.f;
    for (int px = p0; px < p1; ++px) {
        const int y = px / w, x = px - y * w;
        const float dcv = gS[(1 + px) * BOTT + c];
        sdc += dcv;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int yy = y + i - 3;
            const bool vy = yy >= 0 && yy < h;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int xx = x + j - 3;
                const bool ok = vy && xx >= 0 && xx < w;
                const float tr = tS[(ok ? 1 + yy * w + xx : 0) * BOTT + c];
                const float tv = ok ? tr : 0.f;
                dkm[i * 7 + j] = fmaf(dcv, tv, dkm[i * 7 + j]);
                s3 = fmaf(k3[i * 7 + j], tv, s3);
                if (i >= 1 && i <= 5 && j >= 1 && j <= 5) s2 = fmaf(k2[(i - 1) * 5 + (j - 1)], tv, s2);
                if (i >= 2 && i <= 4 && j >= 2 && j <= 4) s1 = fmaf(k1[(i - 2) * 3 + (j - 2)], tv, s1);
            }
        }
        dwm1 = fmaf(dcv, fmaf(f, s1, b1), dwm1);     // Σ dc·conv_k  (conv_k = f·(k_k*t) + b_k)
        dwm2 = fmaf(dcv, fmaf(f, s2, b2), dwm2);
        dwm3 = fmaf(dcv, fmaf(f, s3, b3), dwm3);
    }
    // reduce over the pixel groups with LDS atomics into [64][50] (the c tile is free now)
    float* red = cS;
    for (int i = tid; i < RED_FLOATS; i += 512) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 49; ++i) atomicAdd(red + c * 50 + i, dkm[i]);
    atomicAdd(red + c * 50 + 49, sdc);
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 49; ++i) dkm[i] = f * red[c * 50 + i];      // Σ dc·xf[nbr]
        sdc = red[c * 50 + 49];
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                grad_out(wsrow, WS_C3W, p.g_conv3_w, c * 49 + i * 7 + j, w3 * dkm[i * 7 + j]);
                if (i >= 1 && i <= 5 && j >= 1 && j <= 5) grad_out(wsrow, WS_C2W, p.g_conv2_w, c * 25 + (i - 1) * 5 + (j - 1), w2 * dkm[i * 7 + j]);
                if (i >= 2 && i <= 4 && j >= 2 && j <= 4) grad_out(wsrow, WS_C1W, p.g_conv1_w, c * 9 + (i - 2) * 3 + (j - 2), w1 * dkm[i * 7 + j]);
            }
        grad_out(wsrow, WS_C1B, p.g_conv1_b, c, w1 * sdc);
        grad_out(wsrow, WS_C2B, p.g_conv2_b, c, w2 * sdc);
        grad_out(wsrow, WS_C3B, p.g_conv3_b, c, w3 * sdc);
    }
    float dpool_c = 0.f;
    if (has_noise) {
        // block-wide sums of dwm_k over channels and pixel groups
        __syncthreads();
        float r1 = wave_sum(dwm1), r2 = wave_sum(dwm2), r3 = wave_sum(dwm3);
        if (c == 0) { scr[SCR_RED + grp * 4] = r1; scr[SCR_RED + grp * 4 + 1] = r2; scr[SCR_RED + grp * 4 + 2] = r3; }
        __syncthreads();
        if (tid < 16) {
            // softmax backward → logits → hidden (each of the 16 threads recomputes the 3-vector)
            float dwv[3], wv[3] = {w1, w2, w3}, dl[3];
            for (int k = 0; k < 3; ++k) {
                dwv[k] = 0.f;
                for (int q = 0; q < NGRP; ++q) dwv[k] += scr[SCR_RED + 4 * q + k];
            }
            const float dot = dwv[0] * wv[0] + dwv[1] * wv[1] + dwv[2] * wv[2];
            for (int k = 0; k < 3; ++k) dl[k] = wv[k] * (dwv[k] - dot);
            const float hid = scr[SCR_HID + tid], hpre = scr[SCR_HPRE + tid];
            float dh = 0.f;
            for (int k = 0; k < 3; ++k) {
                grad_out(wsrow, WS_NE3W, p.g_ne3_w, k * 16 + tid, dl[k] * hid);
                dh = fmaf(dl[k], p.ne3_w[k * 16 + tid], dh);
            }
            if (tid < 3) grad_out(wsrow, WS_NE3B, p.g_ne3_b, tid, dl[tid]);
            dh = hpre > 0.f ? dh : 0.f;
            scr[SCR_HPRE + tid] = dh;                       // reuse as dh
            grad_out(wsrow, WS_NE1B, p.g_ne1_b, tid, dh);
        }
        __syncthreads();
        if (tid < 64) {
            float dp = 0.f;
            for (int j = 0; j < 16; ++j) dp = fmaf(scr[SCR_HPRE + j], p.ne1_w[j * 64 + tid], dp);
            scr[SCR_DPOOL + tid] = dp / hw;                 // gradient reaching every xf[c][pix] through the pool
        }
        for (int i = tid; i < 16 * 64; i += 512) grad_out(wsrow, WS_NE1W, p.g_ne1_w, i, scr[SCR_HPRE + (i >> 6)] * scr[SCR_POOL + (i & 63)]);
        __syncthreads();
        dpool_c = scr[SCR_DPOOL + c];
    }
    // ---- pass B: dxf = Kᵀ ⋆ dc (+ pool path);  dt = dc + f·dxf;  df = Σ dxf·t
    float dfc = 0.f;
    T* dt = (T*)p.dt;
    for (int px = p0; px < p1; ++px) {
        const int y = px / w, x = px - y * w;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int yy = y - (i - 3);
            const bool vy = yy >= 0 && yy < h;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int xx = x - (j - 3);
                const bool ok = vy && xx >= 0 && xx < w;
                const float gv = gS[(ok ? 1 + yy * w + xx : 0) * BOTT + c];
                acc = fmaf(KM(i, j), ok ? gv : 0.f, acc);
            }
        }
        const float dxf = acc + dpool_c;
        dfc = fmaf(dxf, tS[(1 + px) * BOTT + c], dfc);
        dt[(tok0 + 1 + px) * BOTT + c] = from_f32<T>(fmaf(f, dxf, gS[(1 + px) * BOTT + c]));
    }
    if (grp == 0) dt[tok0 * BOTT + c] = from_f32<T>(gS[c]);
    if (has_freq) {
        __syncthreads();
        scr[SCR_PART + grp * 64 + c] = dfc;
        __syncthreads();
        if (tid < 64) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < NGRP; ++q) a += scr[SCR_PART + q * 64 + tid];
            grad_out(wsrow, WS_FREQ, p.g_freq, tid, a);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 fast path of the spatial op (grid width W known at compile time: 14 for ViT-B/16 at 224, 4 in the unit tests).
// Same arithmetic, same LDS residency, restructured for the machine:
//   * 7×7 stencils (forward, transposed, and the weight-gradient correlation) are ROW STRIPS: a wave owns image rows
//     y = wave, wave+8, …; a lane owns one channel and holds the W outputs of the row in registers; each source row is
//     read from LDS once (W loads) into a zero-padded register strip and feeds 7·W FMAs with compile-time indices.
//     (The per-pixel form spent ~4 VALU instructions of bounds logic and one LDS load per FMA.)
//   * the 1×1 projector, its data gradient and its weight gradient run on the matrix cores
//     (v_mfma_f32_16x16x32_bf16, operands converted from the fp32 LDS tiles on the way into the registers).
//     c and dz tiles use a row stride of 68 floats so that row-per-lane fragment reads are conflict-free.
template <int W, bool BWD>
__global__ __launch_bounds__(512) void mona_spatial_fast_kernel(const uia_mona_spatial_desc p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = p.h, hw = h * W, ntok = hw + 1;
    float* tS = (float*)smem;                                   // [ntok][64]      t (token 0 = CLS)
    float* cS = tS + ntok * BOTT;                               // [hw][FLD]       c = conv + identity  (later: reduction scratch)
    const int cs_floats = (hw * FLD > FAST_RED) ? hw * FLD : FAST_RED;
    float* gS = cS + cs_floats;                                 // [ntok][FLD]     dz, then dc (backward only)
    float* scr = BWD ? gS + ((ntok * FLD > KW_FLOATS) ? ntok * FLD : KW_FLOATS) : cS + cs_floats;
    const int tid = threadIdx.x, lane = tid & 63, c = lane, grp = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const bool has_freq = p.variant == UIA_MONA_FREQ_ENHANCED || p.variant == UIA_MONA_HYBRID;
    const bool has_noise = p.variant == UIA_MONA_NOISE_AWARE || p.variant == UIA_MONA_HYBRID;
    const int ppg = (hw + NGRP - 1) / NGRP;
    const size_t tok0 = (size_t)b * ntok;
    float* wsrow = (BWD && p.ws) ? p.ws + (size_t)b * WS_ROW : nullptr;

    // stencil weights: coalesced 16-byte loads into LDS, then each lane picks its channel's 9 + 25 + 49 taps (odd strides: conflict-free).
    // (Per-lane global loads at a 36 / 100 / 196-byte lane stride touched 64 cache lines per instruction, 83 times.)
    float* wst = BWD ? gS : scr + SCR_SIZE;                     // backward: the dz tile is not written before the projector phase
    for (int i = tid; i < KW_FLOATS / 4; i += 512) {
        const float* src = i < 144 ? p.conv1_w + 4 * i : (i < 544 ? p.conv2_w + 4 * (i - 144) : p.conv3_w + 4 * (i - 544));
        *(f32x4*)(wst + 4 * i) = load4(src);
    }
    load_tokens((const T*)p.t + tok0 * BOTT, tS, ntok, tid);
    __syncthreads();
#if defined(MSB_STOP) && MSB_STOP == 0
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    const float f = has_freq ? p.freq[c] : 1.0f;
    float w1 = 1.f / 3.f, w2 = 1.f / 3.f, w3 = 1.f / 3.f;
    if (has_noise) {
        noise_forward(p, tS, f, hw, ppg, c, grp, tid, scr);
        w1 = scr[SCR_W]; w2 = scr[SCR_W + 1]; w3 = scr[SCR_W + 2];
    }
    float km[49];
    {
        float k1[9], k2[25], k3[49];
#pragma unroll
        for (int i = 0; i < 9; ++i) k1[i] = wst[c * 9 + i];
#pragma unroll
        for (int i = 0; i < 25; ++i) k2[i] = wst[576 + c * 25 + i];
#pragma unroll
        for (int i = 0; i < 49; ++i) k3[i] = wst[2176 + c * 49 + i];
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) km[i * 7 + j] = KM(i, j);
    }
    const float b1 = p.conv1_b[c], b2 = p.conv2_b[c], b3 = p.conv3_b[c];
    const float bm = w1 * b1 + w2 * b2 + w3 * b3;
#if defined(MSB_STOP) && MSB_STOP == 1
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // ---- c[px][ch] = f·(K ⋆ t) + Σ w_k b_k + t
    for (int y = grp; y < h; y += NGRP) {
        float acc[W];
#pragma unroll
        for (int x = 0; x < W; ++x) acc[x] = 0.f;
        stencil_row<W, false>(tS + BOTT, BOTT, h, y, c, km, acc);
#pragma unroll
        for (int x = 0; x < W; ++x) cS[(y * W + x) * FLD + c] = fmaf(f, acc[x], bm) + tS[(1 + y * W + x) * BOTT + c];
    }
    __syncthreads();

#if defined(MSB_STOP) && MSB_STOP == 2
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // ---- projector on the matrix cores: z[px][co] = c[px][co] + Σ_ci c[px][ci]·P[co][ci] + pb[co]
    const float inv_keep = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t thresh = p.p_drop > 0.f ? (uint32_t)fminf(p.p_drop * 4294967296.0f, 4294967295.0f) : 0u;
    auto keep_scale = [&](int tok, int ch) -> float {
        const size_t idx = (tok0 + tok) * BOTT + ch;
        if (p.keep_mask) return p.keep_mask[idx] ? inv_keep : 0.f;
        if (p.p_drop > 0.f) return dropout_keep(p.seed, (uint32_t)idx, thresh) ? inv_keep : 0.f;
        return 1.0f;
    };
    const int mtiles = (hw + 15) >> 4;
    {
        bf16x8 pw[4][2];                                        // B fragments: P rows co = 16nt + li, k = ci = 32kk + 8g .. +7
        float pb[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float* src = p.proj_w + (16 * nt + li) * 64 + 32 * kk + 8 * g;
                pw[nt][kk] = pack8(load4(src), load4(src + 4));
            }
            pb[nt] = p.proj_b[16 * nt + li];
        }
        for (int mt = grp; mt < mtiles; mt += NGRP) {
            int arow = 16 * mt + li;
            arow = arow < hw ? arow : hw - 1;
            bf16x8 af[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float* src = cS + arow * FLD + 32 * kk + 8 * g;
                af[kk] = pack8(*(const f32x4*)src, *(const f32x4*)(src + 4));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], pw[nt][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], pw[nt][1], acc, 0, 0, 0);
                const int co = 16 * nt + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = 16 * mt + 4 * g + r;
                    if (px < hw) {
                        const float z = cS[px * FLD + co] + acc[r] + pb[nt];
                        const float ks = keep_scale(1 + px, co);
                        if (!BWD) ((T*)p.d)[(tok0 + 1 + px) * BOTT + co] = (T)(gelu_erf(z) * ks);
                        else gS[(1 + px) * FLD + co] = to_f32(((const T*)p.dd)[(tok0 + 1 + px) * BOTT + co]) * ks * dgelu_erf(z);    // dz
                    }
                }
            }
        }
        if (grp == NGRP - 1) {   // CLS token bypasses the spatial op (mona.py:132,139)
            const float z = tS[c], ks = keep_scale(0, c);
            if (!BWD) ((T*)p.d)[tok0 * BOTT + c] = (T)(gelu_erf(z) * ks);
            else gS[c] = to_f32(((const T*)p.dd)[tok0 * BOTT + c]) * ks * dgelu_erf(z);
        }
    }
    if (!BWD) return;

#if defined(MSB_STOP) && MSB_STOP == 3
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // ======================================================================== backward
    __syncthreads();
    // ---- dP[co][ci] = Σ_px dz[px][co]·c[px][ci] on the matrix cores (k = pixels, zero beyond hw); db_p[co] = Σ_px dz[px][co]
    {
        const int mt = grp & 3, nt0 = 2 * (grp >> 2);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const int ksteps = (hw + 31) >> 5;
        for (int ks = 0; ks < ksteps; ++ks) {
            bf16x8 a, bb[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int px = 32 * ks + 8 * g + e;
                const int pc = px < hw ? px : hw - 1;
                const float av = gS[(1 + pc) * FLD + 16 * mt + li];
                a[e] = (bf16_t)(px < hw ? av : 0.f);
                bb[0][e] = (bf16_t)cS[pc * FLD + 16 * nt0 + li];
                bb[1][e] = (bf16_t)cS[pc * FLD + 16 * (nt0 + 1) + li];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb[1], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) grad_out(wsrow, WS_PROJ_W, p.g_proj_w, (16 * mt + 4 * g + r) * 64 + 16 * (nt0 + q) + li, acc[q][r]);
        {   // partial of db_p over this wave's share of the pixels (summed in the output phase below)
            float sb = 0.f;
            for (int px = grp; px < hw; px += NGRP) sb += gS[(1 + px) * FLD + c];
            scr[SCR_PART + grp * 64 + c] = sb;
        }
    }
    __syncthreads();
#if defined(MSB_STOP) && MSB_STOP == 4
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // ---- dc = dz + dz·P, in place (the rows of an m-tile belong to one wave)
    {
        bf16x8 pt[4][2];                                        // B fragments: k = co = 32kk + 8g + e, n = ci = 16nt + li  →  P[co][ci]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) pt[nt][kk][e] = (bf16_t)p.proj_w[(32 * kk + 8 * g + e) * 64 + 16 * nt + li];
        for (int mt = grp; mt < mtiles; mt += NGRP) {
            int arow = 16 * mt + li;
            arow = arow < hw ? arow : hw - 1;
            bf16x8 af[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float* src = gS + (1 + arow) * FLD + 32 * kk + 8 * g;
                af[kk] = pack8(*(const f32x4*)src, *(const f32x4*)(src + 4));
            }
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], pt[nt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], pt[nt][1], acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = 16 * mt + 4 * g + r;
                    if (px < hw) gS[(1 + px) * FLD + 16 * nt + li] += acc[nt][r];
                }
        }
    }
    __syncthreads();
#if defined(MSB_STOP) && MSB_STOP == 5
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // ---- pass A: stencil weight gradients dK[i][j] = Σ_px dc[px]·t[px + off]  (and the mixing-weight gradients of the noise variants)
    float dkm[49];
#pragma unroll
    for (int i = 0; i < 49; ++i) dkm[i] = 0.f;
    float sdc = 0.f, dwm1 = 0.f, dwm2 = 0.f, dwm3 = 0.f;
    for (int y = grp; y < h; y += NGRP) {
        float dcr[W];
#pragma unroll
        for (int x = 0; x < W; ++x) { dcr[x] = gS[(1 + y * W + x) * FLD + c]; sdc += dcr[x]; }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int yy = y + i - 3;
            if (yy < 0 || yy >= h) continue;
            float row[W + 6];
#pragma unroll
            for (int x = 0; x < W + 6; ++x) row[x] = 0.f;
#pragma unroll
            for (int x = 0; x < W; ++x) row[3 + x] = tS[(1 + yy * W + x) * BOTT + c];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                float a = 0.f;
#pragma unroll
                for (int x = 0; x < W; ++x) a = fmaf(dcr[x], row[x + j], a);
                dkm[i * 7 + j] += a;
            }
        }
    }
    if (has_noise) {
        // Σ_px dc·conv_k = f·Σ_taps k_k[tap]·dK[tap] (restricted to the sub-stencil) + b_k·Σ dc   — no second sweep needed
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                s3 = fmaf(p.conv3_w[c * 49 + i * 7 + j], dkm[i * 7 + j], s3);
                if (i >= 1 && i <= 5 && j >= 1 && j <= 5) s2 = fmaf(p.conv2_w[c * 25 + (i - 1) * 5 + (j - 1)], dkm[i * 7 + j], s2);
                if (i >= 2 && i <= 4 && j >= 2 && j <= 4) s1 = fmaf(p.conv1_w[c * 9 + (i - 2) * 3 + (j - 2)], dkm[i * 7 + j], s1);
            }
        dwm1 = fmaf(f, s1, b1 * sdc);
        dwm2 = fmaf(f, s2, b2 * sdc);
        dwm3 = fmaf(f, s3, b3 * sdc);
    }
#if defined(MSB_STOP) && MSB_STOP == 6
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // reduce over the eight waves through the (now free) c tile: waves 4-7 park their partials, waves 0-3 add their own on top, and the
    // output phase sums the four [64][51] images.  (LDS float atomics from 8 waves onto the same 3200 addresses took 37 us here.)
    float* red = cS;
    constexpr int RST = 51, RW = 64 * RST;
    if (grp >= 4) {
        float* dst = red + (grp - 4) * RW + c * RST;
#pragma unroll
        for (int i = 0; i < 49; ++i) dst[i] = dkm[i];
        dst[49] = sdc;
    }
    __syncthreads();
    if (grp < 4) {
        float* dst = red + grp * RW + c * RST;
#pragma unroll
        for (int i = 0; i < 49; ++i) dst[i] += dkm[i];
        dst[49] += sdc;
    }
    __syncthreads();
    auto red4 = [&](int ch, int k) { const float* r = red + ch * RST + k; return (r[0] + r[RW]) + (r[2 * RW] + r[3 * RW]); };
    // All 512 threads write the per-image partials in the order of the parameter tensors (consecutive lanes → consecutive floats).
    // The first version had wave 0 write 86 values per lane at a 196-byte lane stride: 42 of the kernel's 82 us (tools/msb_variants.sh).
    for (int i = tid; i < 64 * 49; i += 512) {
        const int ch = i / 49, k = i - 49 * ch;
        grad_out(wsrow, WS_C3W, p.g_conv3_w, i, w3 * (has_freq ? p.freq[ch] : 1.0f) * red4(ch, k));      // f·Σ dc·t[nbr]
    }
    for (int i = tid; i < 64 * 25; i += 512) {
        const int ch = i / 25, k = i - 25 * ch, ki = k / 5, kj = k - 5 * ki;
        grad_out(wsrow, WS_C2W, p.g_conv2_w, i, w2 * (has_freq ? p.freq[ch] : 1.0f) * red4(ch, (ki + 1) * 7 + kj + 1));
    }
    for (int i = tid; i < 64 * 9; i += 512) {
        const int ch = i / 9, k = i - 9 * ch, ki = k / 3, kj = k - 3 * ki;
        grad_out(wsrow, WS_C1W, p.g_conv1_w, i, w1 * (has_freq ? p.freq[ch] : 1.0f) * red4(ch, (ki + 2) * 7 + kj + 2));
    }
    if (tid < 64) {
        const float sd = red4(tid, 49);
        grad_out(wsrow, WS_C1B, p.g_conv1_b, tid, w1 * sd);
        grad_out(wsrow, WS_C2B, p.g_conv2_b, tid, w2 * sd);
        grad_out(wsrow, WS_C3B, p.g_conv3_b, tid, w3 * sd);
    } else if (tid < 128) {                                   // db_p[co] = Σ_px dz[px][co]: the eight per-wave partials of the dz phase
        float sb = 0.f;
#pragma unroll
        for (int q = 0; q < NGRP; ++q) sb += scr[SCR_PART + q * 64 + tid - 64];
        grad_out(wsrow, WS_PROJ_B, p.g_proj_b, tid - 64, sb);
    }
#if defined(MSB_STOP) && MSB_STOP == 7
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    float dpool_c = 0.f;
    if (has_noise) {
        __syncthreads();
        float r1 = wave_sum(dwm1), r2 = wave_sum(dwm2), r3 = wave_sum(dwm3);
        if (c == 0) { scr[SCR_RED + grp * 4] = r1; scr[SCR_RED + grp * 4 + 1] = r2; scr[SCR_RED + grp * 4 + 2] = r3; }
        __syncthreads();
        if (tid < 16) {
            float dwv[3], wv[3] = {w1, w2, w3}, dl[3];
            for (int k = 0; k < 3; ++k) {
                dwv[k] = 0.f;
                for (int q = 0; q < NGRP; ++q) dwv[k] += scr[SCR_RED + 4 * q + k];
            }
            const float dot = dwv[0] * wv[0] + dwv[1] * wv[1] + dwv[2] * wv[2];
            for (int k = 0; k < 3; ++k) dl[k] = wv[k] * (dwv[k] - dot);
            const float hid = scr[SCR_HID + tid], hpre = scr[SCR_HPRE + tid];
            float dh = 0.f;
            for (int k = 0; k < 3; ++k) {
                grad_out(wsrow, WS_NE3W, p.g_ne3_w, k * 16 + tid, dl[k] * hid);
                dh = fmaf(dl[k], p.ne3_w[k * 16 + tid], dh);
            }
            if (tid < 3) grad_out(wsrow, WS_NE3B, p.g_ne3_b, tid, dl[tid]);
            dh = hpre > 0.f ? dh : 0.f;
            scr[SCR_HPRE + tid] = dh;
            grad_out(wsrow, WS_NE1B, p.g_ne1_b, tid, dh);
        }
        __syncthreads();
        if (tid < 64) {
            float dp = 0.f;
            for (int j = 0; j < 16; ++j) dp = fmaf(scr[SCR_HPRE + j], p.ne1_w[j * 64 + tid], dp);
            scr[SCR_DPOOL + tid] = dp / hw;
        }
        for (int i = tid; i < 16 * 64; i += 512) grad_out(wsrow, WS_NE1W, p.g_ne1_w, i, scr[SCR_HPRE + (i >> 6)] * scr[SCR_POOL + (i & 63)]);
        __syncthreads();
        dpool_c = scr[SCR_DPOOL + c];
    }
#if defined(MSB_STOP) && MSB_STOP == 8
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    // ---- pass B: dxf = Kᵀ ⋆ dc (+ pool path);  dt = dc + f·dxf;  df = Σ dxf·t
    float dfc = 0.f;
    T* dt = (T*)p.dt;
    for (int y = grp; y < h; y += NGRP) {
        float acc[W];
#pragma unroll
        for (int x = 0; x < W; ++x) acc[x] = 0.f;
        stencil_row<W, true>(gS + FLD, FLD, h, y, c, km, acc);
#pragma unroll
        for (int x = 0; x < W; ++x) {
            const int px = y * W + x;
            const float dxf = acc[x] + dpool_c;
            dfc = fmaf(dxf, tS[(1 + px) * BOTT + c], dfc);
            dt[(tok0 + 1 + px) * BOTT + c] = (T)fmaf(f, dxf, gS[(1 + px) * FLD + c]);
        }
    }
    if (grp == NGRP - 1) dt[tok0 * BOTT + c] = (T)gS[c];
#if defined(MSB_STOP) && MSB_STOP == 9
    if (BWD && p.B > 0) return;             // diagnostic (tools/msb_variants.sh): time up to here
#endif
    if (has_freq) {
        __syncthreads();
        scr[SCR_PART + grp * 64 + c] = dfc;
        __syncthreads();
        if (tid < 64) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < NGRP; ++q) a += scr[SCR_PART + q * 64 + tid];
            grad_out(wsrow, WS_FREQ, p.g_freq, tid, a);
        }
    }
}

// g[param][i] += Σ_b ws[b][off + i]   (fixed summation order inside a launch; the final add into the gradient buffer is atomic)
// 64 columns per block × 16 image phases: every phase walks its images with 4 independent loads in flight (16 loads per thread at
// B = 256; the four-phase version had 64 dependent-issue loads per thread and took 17 us for 11 MB).
constexpr int WSR_PH = 16;
__global__ __launch_bounds__(64 * WSR_PH) void mona_ws_reduce_kernel(int B, const float* __restrict__ ws, uia_mona_spatial_desc p, int has_freq, int has_noise) {
    __shared__ float part[WSR_PH][64];
    const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (col < WS_ROW) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int b = ph;
        for (; b + 3 * WSR_PH < B; b += 4 * WSR_PH) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += ws[(size_t)(b + WSR_PH * k) * WS_ROW + col];
        }
        for (; b < B; b += WSR_PH) a[0] += ws[(size_t)b * WS_ROW + col];
        s = (a[0] + a[1]) + (a[2] + a[3]);
    }
    part[ph][cx] = s;
    __syncthreads();
    if (ph != 0 || col >= WS_ROW) return;
    s = 0.f;
#pragma unroll
    for (int q = 0; q < WSR_PH; q += 4) s += (part[q][cx] + part[q + 1][cx]) + (part[q + 2][cx] + part[q + 3][cx]);
    float* dst = nullptr;
    int idx = 0;
    if (col < WS_PROJ_B) { dst = p.g_proj_w; idx = col - WS_PROJ_W; }
    else if (col < WS_C1W) { dst = p.g_proj_b; idx = col - WS_PROJ_B; }
    else if (col < WS_C1B) { dst = p.g_conv1_w; idx = col - WS_C1W; }
    else if (col < WS_C2W) { dst = p.g_conv1_b; idx = col - WS_C1B; }
    else if (col < WS_C2B) { dst = p.g_conv2_w; idx = col - WS_C2W; }
    else if (col < WS_C3W) { dst = p.g_conv2_b; idx = col - WS_C2B; }
    else if (col < WS_C3B) { dst = p.g_conv3_w; idx = col - WS_C3W; }
    else if (col < WS_FREQ) { dst = p.g_conv3_b; idx = col - WS_C3B; }
    else if (col < WS_NE1W) { if (!has_freq) return; dst = p.g_freq; idx = col - WS_FREQ; }
    else if (!has_noise) return;
    else if (col < WS_NE1B) { dst = p.g_ne1_w; idx = col - WS_NE1W; }
    else if (col < WS_NE3W) { dst = p.g_ne1_b; idx = col - WS_NE1B; }
    else if (col < WS_NE3B) { dst = p.g_ne3_w; idx = col - WS_NE3W; if (idx >= 48) return; }
    else { dst = p.g_ne3_b; idx = col - WS_NE3B; if (idx >= 3) return; }
    atomicAdd(dst + idx, s);          // one add per element and launch; atomic because two micro-batch streams may reduce into the same gradient buffer
}

size_t spatial_lds(int hw, bool bwd) {
    const size_t tile = (size_t)(hw + 1) * BOTT;
    const size_t cs = bwd ? (tile > (size_t)RED_FLOATS ? tile : (size_t)RED_FLOATS) : tile;
    return (tile * (bwd ? 2 : 1) + cs + SCR_SIZE) * sizeof(float);
}

template <int W, bool BWD>
int launch_spatial_fast(hipStream_t stream, const uia_mona_spatial_desc& p) {
    auto kern = mona_spatial_fast_kernel<W, BWD>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(p.B), dim3(512), spatial_fast_lds(p.h * W, BWD), stream, p);
    return 0;
}

template <typename T, bool BWD>
int launch_spatial(hipStream_t stream, const uia_mona_spatial_desc& p) {
    const size_t lds = spatial_lds(p.h * p.w, BWD);
    bool fast = false;
    if (sizeof(T) == 2 && (p.w == 14 || p.w == 4) && spatial_fast_lds(p.h * p.w, BWD) <= 160 * 1024) {
        static int allow = -1;                       // UIA_MONA_FAST=0 selects the per-pixel kernel (cross-checks)
        if (allow < 0) { const char* e = getenv("UIA_MONA_FAST"); allow = (e && e[0] == '0') ? 0 : 1; }
        fast = allow != 0;
    }
    if (fast) {
        if (p.w == 14) launch_spatial_fast<14, BWD>(stream, p);
        else launch_spatial_fast<4, BWD>(stream, p);
    } else {
    auto kern = mona_spatial_kernel<T, BWD>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(p.B), dim3(512), lds, stream, p);
    }
    if (BWD && p.ws) {
        const bool has_freq = p.variant == UIA_MONA_FREQ_ENHANCED || p.variant == UIA_MONA_HYBRID;
        const bool has_noise = p.variant == UIA_MONA_NOISE_AWARE || p.variant == UIA_MONA_HYBRID;
        const int cols = has_noise ? WS_ROW : (has_freq ? WS_NE1W : WS_FREQ);     // the noise-estimator / frequency columns only where the variant has them
        hipLaunchKernelGGL(mona_ws_reduce_kernel, dim3((cols + 63) / 64), dim3(64 * WSR_PH), 0, stream, p.B, p.ws, p, (int)has_freq, (int)has_noise);
    }
    UIA_CHECK_LAUNCH();
    return 0;
}

int check_spatial(const uia_mona_spatial_desc& p, bool bwd) {
    UIA_CHECK_ARG(p.variant >= 0 && p.variant <= 3, "uia_mona_spatial: bad variant %d", p.variant);
    UIA_CHECK_ARG(p.bott == BOTT, "uia_mona_spatial: bottleneck %d unsupported (64 only)", p.bott);
    UIA_CHECK_ARG(p.B > 0 && p.h > 0 && p.w > 0, "uia_mona_spatial: empty problem");
    UIA_CHECK_ARG(spatial_lds(p.h * p.w, bwd) <= 160 * 1024, "uia_mona_spatial: %dx%d grid does not fit the 160 KiB LDS tile", p.h, p.w);
    UIA_CHECK_ARG(p.t && p.conv1_w && p.conv1_b && p.conv2_w && p.conv2_b && p.conv3_w && p.conv3_b && p.proj_w && p.proj_b, "uia_mona_spatial: null parameter");
    const bool has_freq = p.variant == UIA_MONA_FREQ_ENHANCED || p.variant == UIA_MONA_HYBRID;
    const bool has_noise = p.variant == UIA_MONA_NOISE_AWARE || p.variant == UIA_MONA_HYBRID;
    UIA_CHECK_ARG(!has_freq || p.freq, "uia_mona_spatial: variant needs freq_filter");
    UIA_CHECK_ARG(!has_noise || (p.ne1_w && p.ne1_b && p.ne3_w && p.ne3_b), "uia_mona_spatial: variant needs noise_estimator parameters");
    UIA_CHECK_ARG(p.p_drop >= 0.f && p.p_drop < 1.f, "uia_mona_spatial: p_drop %f", p.p_drop);
    UIA_CHECK_ARG((((uintptr_t)p.conv1_w | (uintptr_t)p.conv2_w | (uintptr_t)p.conv3_w | (uintptr_t)p.proj_w) & 15) == 0,
                  "uia_mona_spatial: conv / projector weights must be 16-byte aligned");
    if (!bwd) { UIA_CHECK_ARG(p.d, "uia_mona_spatial_fwd: null output"); }
    else {
        UIA_CHECK_ARG(p.dd && p.dt && p.g_conv1_w && p.g_conv1_b && p.g_conv2_w && p.g_conv2_b && p.g_conv3_w && p.g_conv3_b && p.g_proj_w && p.g_proj_b,
                      "uia_mona_spatial_bwd: null gradient buffer");
        UIA_CHECK_ARG(!has_freq || p.g_freq, "uia_mona_spatial_bwd: variant needs g_freq");
        UIA_CHECK_ARG(!has_noise || (p.g_ne1_w && p.g_ne1_b && p.g_ne3_w && p.g_ne3_b), "uia_mona_spatial_bwd: variant needs noise_estimator gradient buffers");
    }
    return 0;
}

}  // namespace

int uia_mona_pre_fwd_t_launch(hipStream_t stream, int dtype, int M, int D, const float* x, const float* nw, const float* nb, const float* gamma,
                              const float* gammax, float eps, void* u, const void* w1, long ldw1, const float* b1, void* t, long ldt) {
    UIA_CHECK_ARG(dtype == UIA_BF16 && D == 768 && M > 0, "uia_mona_pre_fwd_t: bf16, D = 768 (got dtype %d, D = %d)", dtype, D);
    UIA_CHECK_ARG(x && nw && nb && gamma && gammax && u && w1 && t && ldw1 >= D && ldw1 % 8 == 0 && ldt >= 64 && ldt % 4 == 0 && (uintptr_t)w1 % 16 == 0 && (uintptr_t)t % 8 == 0 &&
                      (!b1 || (uintptr_t)b1 % 16 == 0),
                  "uia_mona_pre_fwd_t: null tensor, or W1 [64, D] / t [M, 64] rows not 16- / 8-byte aligned");
    constexpr int KS = 24;
    const int lds = 16 * (2 * 32 * KS + 16);
    const int ntiles = (M + 15) / 16;
    int blocks = ntiles;
    int per_cu = 0, dev = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mona_pre_fwd_t_kernel<KS>, 256, lds) == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0 && per_cu * ncu < blocks) blocks = per_cu * ncu;
    hipLaunchKernelGGL(mona_pre_fwd_t_kernel<KS>, dim3(blocks), dim3(256), lds, stream, M, x, nw, nb, gamma, gammax, eps, (bf16_t*)u, (const bf16_t*)w1, ldw1, b1, (bf16_t*)t, ldt);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_mona_pre_fwd_launch(hipStream_t stream, int dtype, int M, int D, const float* x, const float* nw, const float* nb, const float* gamma,
                            const float* gammax, float eps, void* u) {
    UIA_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "uia_mona_pre_fwd: unsupported shape M=%d D=%d", M, D);
    UIA_CHECK_ARG(x && nw && nb && gamma && gammax && u, "uia_mona_pre_fwd: null tensor");
    const dim3 grid((M + 3) / 4), block(256);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(mona_pre_fwd_kernel<bf16_t>, grid, block, 0, stream, M, D, x, nw, nb, gamma, gammax, eps, (bf16_t*)u);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(mona_pre_fwd_kernel<float>, grid, block, 0, stream, M, D, x, nw, nb, gamma, gammax, eps, (float*)u);
    else { uia_set_error("uia_mona_pre_fwd: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

static int mona_pre_bwd_blocks(int M) {
    const int blocks = (M + 3) / 4;
    return blocks > 1024 ? 1024 : blocks;
}

size_t uia_mona_pre_bwd_ws_floats(int M, int D) { return (size_t)mona_pre_bwd_blocks(M) * 4 * (size_t)D; }

int uia_mona_pre_bwd_launch(hipStream_t stream, int dtype, int M, int D, const void* du, const float* x, const float* dy, const float* nw,
                            const float* nb, const float* gamma, const float* gammax, float eps, float* dx32, void* dxT, float* g_gamma,
                            float* g_gammax, float* g_nw, float* g_nb, float* ws, long dxT_kb_rows, const void* dt, long ldt, const void* w1t, long ldw1,
                            const void* dy_hi, const int8_t* dy_lo, int8_t* dx_lo) {
    UIA_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "uia_mona_pre_bwd: unsupported shape M=%d D=%d", M, D);
    UIA_CHECK_ARG(dxT_kb_rows == 0 || (dxT && dxT_kb_rows >= M && (D * (dtype == UIA_BF16 ? 2 : 4)) % 64 == 0 && dxT_kb_rows * (long)D < (1L << 31)),
                  "uia_mona_pre_bwd: dxT_kb_rows=%ld needs dxT, at least M=%d rows and whole 64-byte column blocks", dxT_kb_rows, M);
    const bool fuse = dt != nullptr;                     // du = dt·W1 computed by this launch (uia_mona_pre_bwd_du)
    UIA_CHECK_ARG((du || fuse) && x && nw && nb && gamma && gammax && g_gamma && g_gammax && g_nw && g_nb && ws, "uia_mona_pre_bwd: null tensor");
    UIA_CHECK_ARG(!fuse || (dtype == UIA_BF16 && w1t && D % 64 == 0 && D <= 768 && ldt >= 64 && ldt % 8 == 0 && ldw1 >= 64 && ldw1 % 8 == 0 && (uintptr_t)dt % 16 == 0 && (uintptr_t)w1t % 16 == 0),
                  "uia_mona_pre_bwd_du: bf16, bottleneck 64 (dt [M, 64], W1ᵀ [D, 64], 16-byte aligned rows), D a multiple of 64 up to 768");
    const bool r3 = dy_lo != nullptr || dx_lo != nullptr;
    UIA_CHECK_ARG(!r3 || (fuse && dy_hi && dy_lo && dx_lo && dxT && !dy && !dx32 && (uintptr_t)dy_hi % 8 == 0 && (uintptr_t)dy_lo % 4 == 0 && (uintptr_t)dx_lo % 4 == 0 && (uintptr_t)dxT % 8 == 0),
                  "uia_mona_pre_bwd_du3: three-byte residual gradients need the fused form, dy as (dy_hi, dy_lo) with no fp32 dy, and dx as (dxT, dx_lo) with no dx32");
    UIA_CHECK_ARG(r3 || dy || !(dx32 || dxT), "uia_mona_pre_bwd: dx requested without dy");
    int blocks = mona_pre_bwd_blocks(M);
    const size_t lds = (size_t)8 * D * sizeof(float) + (fuse ? (size_t)16 * D * 2 : 0);   // [2][3][D] reduction area + [2][D] parameter vectors (+ the du tile, [16][D] bf16)
    const int nvsel = D <= 256 ? 1 : (D <= 768 ? 3 : 4);
    // persistent grid: exactly as many workgroups as are resident at once (a second, partial round would leave most CUs idle at the end)
#define UIA_PRE_BWD(TT, NVV) do {                                                                                                              \
        auto kern = dxT_kb_rows ? mona_pre_bwd_kernel<TT, NVV, true> : mona_pre_bwd_kernel<TT, NVV, false>;                                   \
        int per_cu = 0, dev = 0, ncu = 0;                                                                                                      \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) == hipSuccess && per_cu > 0 &&                               \
            hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&       \
            ncu > 0 && per_cu * ncu < blocks) blocks = per_cu * ncu;                                                                           \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, M, D, (const TT*)du, x, dy, nw, gamma, gammax, eps, dx32, (TT*)dxT, ws, dxT_kb_rows, \
                           (const TT*)nullptr, 0l, (const TT*)nullptr, 0l, (const bf16_t*)nullptr, (const int8_t*)nullptr, (int8_t*)nullptr);    \
    } while (0)
    if (fuse) {
        auto kern = r3 ? (dxT_kb_rows ? mona_pre_bwd_kernel<bf16_t, 3, true, true, true> : mona_pre_bwd_kernel<bf16_t, 3, false, true, true>)
                       : (dxT_kb_rows ? mona_pre_bwd_kernel<bf16_t, 3, true, true> : mona_pre_bwd_kernel<bf16_t, 3, false, true>);
        static UiaDevOnce once[4];
        UIA_ENSURE_LDS_ATTR(once[(r3 ? 2 : 0) + (dxT_kb_rows ? 1 : 0)], kern, 160 * 1024);
        int per_cu = 0, dev = 0, ncu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) == hipSuccess && per_cu > 0 && hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0 && per_cu * ncu < blocks) blocks = per_cu * ncu;
        const int ntiles = (M + 15) / 16;
        if (blocks > ntiles) blocks = ntiles;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, M, D, (const bf16_t*)nullptr, x, dy, nw, gamma, gammax, eps, dx32, (bf16_t*)dxT, ws, dxT_kb_rows,
                           (const bf16_t*)dt, ldt, (const bf16_t*)w1t, ldw1, (const bf16_t*)dy_hi, dy_lo, dx_lo);
        UIA_CHECK_LAUNCH();
        hipLaunchKernelGGL(mona_pre_reduce_kernel, dim3((3 * D + 255) / 256, PRE_RED_SPLIT), dim3(256), 0, stream, blocks, D, ws, nw, nb, gamma, g_gamma, g_gammax, g_nw, g_nb);
        UIA_CHECK_LAUNCH();
        return 0;
    }
    if (dtype == UIA_BF16) { if (nvsel == 1) UIA_PRE_BWD(bf16_t, 1); else if (nvsel == 3) UIA_PRE_BWD(bf16_t, 3); else UIA_PRE_BWD(bf16_t, 4); }
    else if (dtype == UIA_F32) { if (nvsel == 1) UIA_PRE_BWD(float, 1); else if (nvsel == 3) UIA_PRE_BWD(float, 3); else UIA_PRE_BWD(float, 4); }
    else { uia_set_error("uia_mona_pre_bwd: bad dtype %d", dtype); return -1; }
#undef UIA_PRE_BWD
    UIA_CHECK_LAUNCH();
    hipLaunchKernelGGL(mona_pre_reduce_kernel, dim3((3 * D + 255) / 256, PRE_RED_SPLIT), dim3(256), 0, stream, blocks, D, ws, nw, nb, gamma, g_gamma, g_gammax, g_nw, g_nb);
    UIA_CHECK_LAUNCH();
    return 0;
}

size_t uia_mona_spatial_ws_floats(int B) { return (size_t)B * WS_ROW; }

int uia_mona_spatial_fwd_launch(hipStream_t stream, int dtype, const uia_mona_spatial_desc& p) {
    if (int rc = check_spatial(p, false)) return rc;
    if (dtype == UIA_BF16) return launch_spatial<bf16_t, false>(stream, p);
    if (dtype == UIA_F32) return launch_spatial<float, false>(stream, p);
    uia_set_error("uia_mona_spatial_fwd: bad dtype %d", dtype);
    return -1;
}

int uia_mona_spatial_bwd_launch(hipStream_t stream, int dtype, const uia_mona_spatial_desc& p) {
    if (int rc = check_spatial(p, true)) return rc;
    if (dtype == UIA_BF16) return launch_spatial<bf16_t, true>(stream, p);
    if (dtype == UIA_F32) return launch_spatial<float, true>(stream, p);
    uia_set_error("uia_mona_spatial_bwd: bad dtype %d", dtype);
    return -1;
}
