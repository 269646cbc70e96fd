// mona_fused.hip — the whole Mona adapter forward of one image in ONE workgroup (bf16 operands, fp32 accumulate):
//
//     y = x + project2( drop( gelu( spatial( project1( LN(x)·γ + x·γx ) ) ) ) )
//
// Reference: /root/reference/src/adapters/mona.py:319-362 (FreqEnhancedMona.forward; :96-151, :198-253, :427-487 for the other variants)
// with the op bodies :75-93, :159-195, :261-295, :370-424.  Equations: SURVEY.md Appendix E.1.
//
// The unfused path runs it as four launches (mona_pre_fwd -> N = 64 stream GEMM -> mona_spatial_fwd -> K = 64 ring GEMM) and moves
// u [M, D], t and d [M, 64] through HBM between them: 722 MB per layer at M = 50 432, D = 768.  Here one 512-thread workgroup owns an
// image (1 + h·w tokens; the spatial stage needs all of them at once) and walks three LDS-staged phases:
//
//   phase 1  t = (LN(x)·γ + x·γx)·W1ᵀ + b1      16-row tiles on the matrix cores (v_mfma_f32_16x16x32_bf16): the u fragment of a lane is
//            built in registers from its 32 bytes of x (two sweeps over the tile: row statistics, then u), W1 (64 x D) is resident in
//            LDS for the image, t lands in an fp32 [tokens][64] LDS tile (it is never rounded to bf16 on its way to the stencils)
//   phase 2  the spatial op of mona.hip's fast path on that tile: f_c scale, merged 7x7 depth-wise stencil as row strips, 1x1 projector on
//            MFMA, exact-erf GELU, dropout -> d, kept in LDS as the bf16 B operand of phase 3 (and written out for the backward)
//   phase 3  y = x + d·W2ᵀ + b2                 every wave owns D/8 output columns with its W2 fragments in registers and sweeps the
//            row tiles; x is re-read for the residual (Infinity Cache: the image's rows were read by this CU a few microseconds ago),
//            y leaves as fp32 + the T copy (row-major or K-blocked) + the row sums (Σ, Σ²) of the LayerNorm folded into the next GEMM
//
// HBM traffic per layer: x twice, y, y_T, (u for the weight gradient), t, d: 632 MB against 722.  One workgroup per CU (160 KB of LDS).
#include "mona_spatial.h"

namespace {
using namespace uia_mona;

constexpr int NTW_MAX = 6;                         // 16-column tiles per wave in phase 3: D <= 8 waves x 6 x 16 = 768
constexpr float ROWSUM_SCALE_F = 1073741824.0f, ROWSUM_PART_MAX_F = 5.0e8f;

__host__ __device__ constexpr int fused_rega_bytes(int D, int hw) {
    return (64 * (2 * D + 16) > (hw * FLD + KW_FLOATS) * 4) ? 64 * (2 * D + 16) : (hw * FLD + KW_FLOATS) * 4;
}
__host__ __device__ constexpr int fused_lds_bytes(int D, int hw) {
    const int mt = (hw + 1 + 15) / 16;
    const int tail = (3 * D > SCR_SIZE ? 3 * D : SCR_SIZE) * 4;
    const int rowacc = 16 * mt * 16;                                        // [rows][2] 64-bit row sums (phase 3), over the statistics + parameter area
    const int regc = 16 * mt * 8 + tail;
    return fused_rega_bytes(D, hw) + (hw + 1) * BOTT * 4 + (regc > rowacc ? regc : rowacc);
}

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int W>
__global__ __launch_bounds__(512) void mona_fused_fwd_kernel(const uia_mona_fused_desc q) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uia_mona_spatial_desc& p = q.sp;
    const int D = q.D, h = p.h, hw = h * W, ntok = hw + 1;
    const int mtiles = (ntok + 15) >> 4;
    const int ldw1 = 2 * D + 16;                                            // bytes per row of the W1 image (+16: staggers the rows over the banks)
    char* regA = smem;                                                      // phase 1: W1 image; phase 2: c tile + stencil-weight staging
    float* tS = (float*)(smem + fused_rega_bytes(D, hw));                   // [ntok][64] fp32 t; phase 2 end / phase 3: the bf16 d tile overlays it
    float* stats = tS + ntok * BOTT;                                        // [16·mtiles][2] (mean, rstd)
    float* prm = stats + 32 * mtiles;                                       // phase 1: [3][D] = γ·w_n, γ·b_n, γx; phase 2: the spatial op's scratch
    const int tid = threadIdx.x, lane = tid & 63, c = lane, grp = tid >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const size_t tok0 = (size_t)b * ntok;
    const bool has_freq = p.variant == UIA_MONA_FREQ_ENHANCED || p.variant == UIA_MONA_HYBRID;
    const bool has_noise = p.variant == UIA_MONA_NOISE_AWARE || p.variant == UIA_MONA_HYBRID;

    // ---------------------------------------------------------------- phase 0: W1 image and the per-column vectors of `pre`
    {
        const int cpr = D >> 3;                                             // 16-byte chunks per row of W1
        for (int i = tid; i < 64 * cpr; i += 512) {
            const int r = i / cpr, cc = i - r * cpr;
            *(uint4*)(regA + r * ldw1 + cc * 16) = *(const uint4*)((const char*)q.w1 + ((size_t)r * D) * 2 + cc * 16);
        }
        for (int i = tid; i < D; i += 512) {
            const float gm = q.gamma[i];
            prm[i] = gm * q.norm_w[i];
            prm[D + i] = gm * q.norm_b[i];
            prm[2 * D + i] = q.gammax[i];
        }
    }
    __syncthreads();
#if defined(MF_STOP) && MF_STOP == 0
    if (p.B > 0) return;                    // diagnostic builds (tools/mff_variants.sh): time up to here
#endif

    // ---------------------------------------------------------------- phase 1: t = u·W1ᵀ + b1, u = x̂·(γ w_n) + γ b_n + x·γx
    {
        const int KS = D >> 5;
        for (int mt = wave; mt < mtiles; mt += 8) {
            const int row = 16 * mt + li;
            const int rc = row < ntok ? row : ntok - 1;
            const float* xrow = q.x + (tok0 + rc) * D;
            const float* xr = xrow + 8 * g;                                 // + 32·kk: the lane's eight columns of k-step kk
            // sweep 1: row statistics, shifted by the row's first element (E[(x−x0)²] − E[x−x0]² keeps its digits for any row mean)
            const float x0 = xrow[0];
            float s1 = 0.f, s2 = 0.f;
            for (int kk = 0; kk < KS; kk += 4) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = kk + i < KS;
                    v[2 * i] = ok ? load4(xr + 32 * (kk + i)) : f32x4{x0, x0, x0, x0};
                    v[2 * i + 1] = ok ? load4(xr + 32 * (kk + i) + 4) : f32x4{x0, x0, x0, x0};
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float dlt = v[i][e] - x0; s1 += dlt; s2 = fmaf(dlt, dlt, s2); }
            }
            s1 = rows_sum(s1);
            s2 = rows_sum(s2);
            const float m1 = s1 / D;
            const float mean = x0 + m1;
            const float rstd = rsqrtf(fmaxf(fmaf(-m1, m1, s2 / D), 0.f) + q.eps);
#if defined(MF_STOP) && MF_STOP == 1
            if (p.B > 0) { if (lane == 0) tS[mt] = mean + rstd; continue; }
#endif
            // sweep 2 (the tile is in L1 / L2 now): u fragments -> MFMA against the resident W1
            f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            T* urow = q.u_out ? (T*)q.u_out + (tok0 + rc) * D + 8 * g : nullptr;
            for (int kk0 = 0; kk0 < KS; kk0 += 4) {
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = kk0 + i < KS;
                    v[2 * i] = ok ? load4(xr + 32 * (kk0 + i)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    v[2 * i + 1] = ok ? load4(xr + 32 * (kk0 + i) + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = kk0 + i;
                    if (kk < KS) {
                        const float* pc = prm + 32 * kk + 8 * g;
                        f32x4 ua, ub;
#pragma unroll
                        for (int hlf = 0; hlf < 2; ++hlf) {
                            const f32x4 wg = *(const f32x4*)(pc + 4 * hlf), bg = *(const f32x4*)(pc + D + 4 * hlf), gx = *(const f32x4*)(pc + 2 * D + 4 * hlf);
                            const f32x4 xv = v[2 * i + hlf];
                            f32x4 uu;
#pragma unroll
                            for (int e = 0; e < 4; ++e) uu[e] = fmaf((xv[e] - mean) * rstd, wg[e], fmaf(xv[e], gx[e], bg[e]));
                            if (hlf == 0) ua = uu; else ub = uu;
                        }
                        const bf16x8 uf = pack8(ua, ub);
                        if (urow && row < ntok) *(bf16x8*)(urow + 32 * kk) = uf;
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const bf16x8 wf = *(const bf16x8*)(regA + (16 * nt + li) * ldw1 + 64 * kk + 16 * g);
                            acc[nt] = mfma16(uf, wf, acc[nt]);
                        }
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float bb = q.b1[16 * nt + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ro = 16 * mt + 4 * g + r;
                    if (ro < ntok) tS[ro * BOTT + 16 * nt + li] = acc[nt][r] + bb;
                }
            }
        }
    }
    __syncthreads();
    // t for the backward (it recomputes c and z from it): T copy, coalesced
    if (q.t_out) {
        T* dst = (T*)q.t_out + tok0 * BOTT;
        for (int i = tid; i < ntok * 8; i += 512) {
            const f32x4 a = *(const f32x4*)(tS + i * 8), bq = *(const f32x4*)(tS + i * 8 + 4);
            *(bf16x8*)(dst + (size_t)i * 8) = pack8(a, bq);
        }
    }

#if defined(MF_STOP) && MF_STOP <= 2
    if (p.B > 0) return;
#endif
    // ---------------------------------------------------------------- phase 2: the spatial op (mona.hip, mona_spatial_fast_kernel<W, false>)
    float* cS = (float*)regA;                                               // [hw][FLD]
    float* wst = cS + hw * FLD;                                             // stencil weights, staged coalesced
    float* scr = prm;
    for (int i = tid; i < KW_FLOATS / 4; i += 512) {
        const float* src = i < 144 ? p.conv1_w + 4 * i : (i < 544 ? p.conv2_w + 4 * (i - 144) : p.conv3_w + 4 * (i - 544));
        *(f32x4*)(wst + 4 * i) = load4(src);
    }
    __syncthreads();
    const int ppg = (hw + NGRP - 1) / NGRP;
    const float f = has_freq ? p.freq[c] : 1.0f;
    float w1 = 1.f / 3.f, w2 = 1.f / 3.f, w3 = 1.f / 3.f;
    if (has_noise) {
        noise_forward(p, tS, f, hw, ppg, c, grp, tid, scr);
        w1 = scr[SCR_W]; w2 = scr[SCR_W + 1]; w3 = scr[SCR_W + 2];
    }
    {
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
        const float bm = w1 * p.conv1_b[c] + w2 * p.conv2_b[c] + w3 * p.conv3_b[c];
        for (int y = grp; y < h; y += NGRP) {
            float acc[W];
#pragma unroll
            for (int x = 0; x < W; ++x) acc[x] = 0.f;
            stencil_row<W, false>(tS + BOTT, BOTT, h, y, c, km, acc);
#pragma unroll
            for (int x = 0; x < W; ++x) cS[(y * W + x) * FLD + c] = fmaf(f, acc[x], bm) + tS[(1 + y * W + x) * BOTT + c];
        }
    }
    const float z_cls = tS[c];                                              // the CLS token bypasses the spatial op (mona.py:132,139)
    __syncthreads();                                                        // c complete; t is dead from here on: d overlays it

    // d tile: [16·mtiles][64] bf16, 16-byte chunk j of row r stored at chunk j ^ ((r >> 1) & 7)  (conflict-free fragment reads in phase 3)
    char* dS = (char*)tS;
    auto d_addr = [&](int r, int col) -> T* { return (T*)(dS + r * 128 + ((((col >> 3) ^ (r >> 1)) & 7) << 4)) + (col & 7); };
    const float inv_keep = p.p_drop > 0.f ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const uint32_t thresh = p.p_drop > 0.f ? (uint32_t)fminf(p.p_drop * 4294967296.0f, 4294967295.0f) : 0u;
    auto keep_scale = [&](int tok, int ch) -> float {
        const size_t idx = (tok0 + tok) * BOTT + ch;
        if (p.keep_mask) return p.keep_mask[idx] ? inv_keep : 0.f;
        if (p.p_drop > 0.f) return dropout_keep(p.seed, (uint32_t)idx, thresh) ? inv_keep : 0.f;
        return 1.0f;
    };
    T* dglob = (T*)p.d;
    {
        const int ptiles = (hw + 15) >> 4;
        bf16x8 pw[4][2];
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
        for (int mt = grp; mt < ptiles; mt += NGRP) {
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
                acc = mfma16(af[0], pw[nt][0], acc);
                acc = mfma16(af[1], pw[nt][1], acc);
                const int co = 16 * nt + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = 16 * mt + 4 * g + r;
                    if (px < hw) {
                        const float z = cS[px * FLD + co] + acc[r] + pb[nt];
                        const T dv = (T)(gelu_erf(z) * keep_scale(1 + px, co));
                        *d_addr(1 + px, co) = dv;
                        if (dglob) dglob[(tok0 + 1 + px) * BOTT + co] = dv;
                    }
                }
            }
        }
        if (grp == NGRP - 1) {
            const T dv = (T)(gelu_erf(z_cls) * keep_scale(0, c));
            *d_addr(0, c) = dv;
            if (dglob) dglob[tok0 * BOTT + c] = dv;
        }
    }
    // phase 3's row-sum accumulators: [16·mtiles][2] 64-bit fixed point (units of 2^-30), over the statistics / scratch area (dead now:
    // the noise estimator's scratch was last read before the stencils)
    unsigned long long* rowacc = (unsigned long long*)stats;
    if (q.rowsum_out)
        for (int i = tid; i < 32 * mtiles; i += 512) rowacc[i] = 0ull;
    __syncthreads();

#if defined(MF_STOP) && MF_STOP <= 3
    if (p.B > 0) return;
#endif
    // ---------------------------------------------------------------- phase 3: y = x + d·W2ᵀ + b2 ; T copy ; row sums
    {
        const int ntw = D >> 7;                                             // 16-column tiles per wave (D / 8 columns)
        const int col0 = wave * (D >> 3);
        uint4 wf[NTW_MAX][2];                                               // MFMA A operand: W2 rows (output columns) of the wave, both k halves
        f32x4 b2v[NTW_MAX];
#pragma unroll
        for (int nt = 0; nt < NTW_MAX; ++nt) {
            if (nt < ntw) {
                const char* src = (const char*)q.w2 + ((size_t)(col0 + 16 * nt + li) * 64 + 8 * g) * 2;
                wf[nt][0] = *(const uint4*)src;
                wf[nt][1] = *(const uint4*)(src + 64);
                b2v[nt] = load4(q.b2 + col0 + 16 * nt + 4 * g);
            } else {
                wf[nt][0] = wf[nt][1] = uint4{0u, 0u, 0u, 0u};
                b2v[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // the lane's residual pieces of a row tile: row 16·mt + li, columns col0 + 16·nt + 4g .. +3.  Requested one tile ahead, BEFORE the
        // previous tile's stores (vmcnt retires in order: a load issued behind stores waits for their acknowledgement)
        f32x4 xa[NTW_MAX], xb[NTW_MAX];
        auto issue_x = [&](int mt, f32x4 (&xv)[NTW_MAX]) {
            const int m = 16 * mt + li;
            const float* src = q.x + (tok0 + (m < ntok ? m : ntok - 1)) * D + col0 + 4 * g;
#pragma unroll
            for (int nt = 0; nt < NTW_MAX; ++nt) xv[nt] = nt < ntw ? load4(src + 16 * nt) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto tile = [&](int mt, f32x4 (&xv)[NTW_MAX]) {
            const int m = 16 * mt + li;
            const int rsw = (m >> 1) & 7;
            const uint4 d0 = *(const uint4*)(dS + m * 128 + ((g ^ rsw) << 4));          // k = 8g .. 8g+7
            const uint4 d1 = *(const uint4*)(dS + m * 128 + (((4 + g) ^ rsw) << 4));    // k = 32 + 8g ..
            float s1 = 0.f, s2 = 0.f;
            const bool ok = m < ntok;
            float* yrow = q.y32 + (tok0 + m) * D + col0 + 4 * g;
#pragma unroll
            for (int nt = 0; nt < NTW_MAX; ++nt) {
                if (nt < ntw) {
                    f32x4 acc = mfma16(__builtin_bit_cast(bf16x8, wf[nt][0]), __builtin_bit_cast(bf16x8, d0), f32x4{0.f, 0.f, 0.f, 0.f});
                    acc = mfma16(__builtin_bit_cast(bf16x8, wf[nt][1]), __builtin_bit_cast(bf16x8, d1), acc);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[e] + b2v[nt][e]) + xv[nt][e];
                    if (ok) {
                        store4(yrow + 16 * nt, v);
                        if (q.yT) {
                            const int n = col0 + 16 * nt + 4 * g;
                            T* dst = q.yT_kb_rows ? (T*)q.yT + ((size_t)(n >> 5) * (size_t)q.yT_kb_rows + tok0 + m) * 32 + (n & 31)
                                                  : (T*)q.yT + (tok0 + m) * D + n;
                            store4(dst, v);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
                    }
                }
            }
            if (q.rowsum_out) {
                s1 = rows_sum(s1);
                s2 = rows_sum(s2);
                if (g == 0 && ok) {
                    if (!(fabsf(s1) < ROWSUM_PART_MAX_F) || !(s2 < ROWSUM_PART_MAX_F)) {      // as uia_gemm's rowsum_out: clamp and flag, never wrap
                        if (q.ln_flag) atomicOr(q.ln_flag, 2);
                        s1 = fminf(fmaxf(s1, -ROWSUM_PART_MAX_F), ROWSUM_PART_MAX_F);
                        s2 = fminf(fmaxf(s2, 0.0f), ROWSUM_PART_MAX_F);
                    }
                    atomicAdd(rowacc + 2 * m, (unsigned long long)llrintf(s1 * ROWSUM_SCALE_F));
                    atomicAdd(rowacc + 2 * m + 1, (unsigned long long)llrintf(s2 * ROWSUM_SCALE_F));
                }
            }
        };
        issue_x(0, xa);
        for (int mt = 0; mt < mtiles; mt += 2) {
            if (mt + 1 < mtiles) issue_x(mt + 1, xb);
            tile(mt, xa);
            if (mt + 1 >= mtiles) break;
            if (mt + 2 < mtiles) issue_x(mt + 2, xa);
            tile(mt + 1, xb);
        }
    }
    if (q.rowsum_out) {
        __syncthreads();
        // this workgroup owns its rows: the sums are stored, not added (the buffer needs no zeroing for this producer)
        for (int i = tid; i < 2 * ntok; i += 512) ((unsigned long long*)q.rowsum_out)[2 * tok0 + i] = rowacc[i];
    }
}

template <int W>
int launch_fused(hipStream_t stream, const uia_mona_fused_desc& q) {
    auto kern = mona_fused_fwd_kernel<W>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(q.sp.B), dim3(512), fused_lds_bytes(q.D, q.sp.h * W), stream, q);
    UIA_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int uia_mona_fused_supported(int dtype, int D, int h, int w, int bott) {
    return dtype == UIA_BF16 && bott == BOTT && (w == 14 || w == 4) && h > 0 && D > 0 && D % 128 == 0 && D <= 128 * NTW_MAX &&
           fused_lds_bytes(D, h * w) <= 160 * 1024;
}

int uia_mona_fused_fwd_launch(hipStream_t stream, int dtype, const uia_mona_fused_desc& q) {
    const uia_mona_spatial_desc& p = q.sp;
    UIA_CHECK_ARG(uia_mona_fused_supported(dtype, q.D, p.h, p.w, p.bott),
                  "uia_mona_fused_fwd: bf16, bottleneck 64, grid width 14 or 4, D a multiple of 128 up to 768 and the image's tiles inside the 160 KiB LDS "
                  "(got dtype %d, D %d, %dx%d, bottleneck %d): use the unfused launches", dtype, q.D, p.h, p.w, p.bott);
    UIA_CHECK_ARG(p.variant >= 0 && p.variant <= 3 && p.B > 0, "uia_mona_fused_fwd: bad variant %d / batch %d", p.variant, p.B);
    UIA_CHECK_ARG(q.x && q.norm_w && q.norm_b && q.gamma && q.gammax && q.w1 && q.b1 && q.w2 && q.b2 && q.y32, "uia_mona_fused_fwd: null tensor");
    UIA_CHECK_ARG(p.conv1_w && p.conv1_b && p.conv2_w && p.conv2_b && p.conv3_w && p.conv3_b && p.proj_w && p.proj_b, "uia_mona_fused_fwd: null parameter");
    const bool has_freq = p.variant == UIA_MONA_FREQ_ENHANCED || p.variant == UIA_MONA_HYBRID;
    const bool has_noise = p.variant == UIA_MONA_NOISE_AWARE || p.variant == UIA_MONA_HYBRID;
    UIA_CHECK_ARG(!has_freq || p.freq, "uia_mona_fused_fwd: variant needs freq_filter");
    UIA_CHECK_ARG(!has_noise || (p.ne1_w && p.ne1_b && p.ne3_w && p.ne3_b), "uia_mona_fused_fwd: variant needs noise_estimator parameters");
    UIA_CHECK_ARG(p.p_drop >= 0.f && p.p_drop < 1.f, "uia_mona_fused_fwd: p_drop %f", p.p_drop);
    UIA_CHECK_ARG((((uintptr_t)q.x | (uintptr_t)q.w1 | (uintptr_t)q.w2 | (uintptr_t)q.y32 | (uintptr_t)q.yT | (uintptr_t)q.u_out | (uintptr_t)q.t_out | (uintptr_t)p.d |
                    (uintptr_t)q.b2 | (uintptr_t)q.rowsum_out | (uintptr_t)p.conv1_w | (uintptr_t)p.conv2_w | (uintptr_t)p.conv3_w | (uintptr_t)p.proj_w) & 15) == 0,
                  "uia_mona_fused_fwd: tensors must be 16-byte aligned");
    const int64_t M = (int64_t)p.B * (p.h * p.w + 1);
    UIA_CHECK_ARG(q.yT_kb_rows == 0 || (q.yT && q.yT_kb_rows >= M), "uia_mona_fused_fwd: yT_kb_rows=%lld needs yT and at least %lld rows", (long long)q.yT_kb_rows, (long long)M);
    return p.w == 14 ? launch_fused<14>(stream, q) : launch_fused<4>(stream, q);
}
