// gemm_epilogue.h — what every GEMM kernel of the library shares: the MFMA wrappers, the activation / LayerNorm-statistics helpers, the LDS-DMA
// issue, and the LDS-staged epilogue (gemm_epilogue_lds) with its compile-time feature masks.  Included by gemm.hip (ring / ping-pong / stream
// kernels) and gemm_quad.hip (four-wave 256 x 256 kernel); everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include "uia_common.h"
#include "uia_kernels.h"

#ifdef UIA_GEMM_STAMPS
static __device__ unsigned long long* uia_stamp_buf = nullptr;   // diagnostic build only (tests/test_gemm_stamps); one copy per translation unit
static __device__ int uia_epi_diag = 0;                          // 1: epilogue without its stores, 2: without stores and operand loads
#define UIA_EPI_STORES ((uia_epi_diag & 3) == 0)
#define UIA_EPI_LOADS ((uia_epi_diag & 3) < 2)
#ifndef UIA_KDIAG
#define UIA_KDIAG 0                                // compile-time: a run-time test inside the K loop disturbs its schedule
#endif
#define UIA_DIAG_NO_FRAGS ((UIA_KDIAG & 4) != 0)   // K loop without its ds_reads
#define UIA_DIAG_NO_DMA ((UIA_KDIAG & 8) != 0)     // K loop without its LDS-DMA (after the prologue)
#define UIA_DIAG_NO_MFMA ((UIA_KDIAG & 16) != 0)   // K loop without its MFMAs
#else
#define UIA_EPI_STORES true
#define UIA_EPI_LOADS true
#define UIA_DIAG_NO_FRAGS false
#define UIA_DIAG_NO_DMA false
#define UIA_DIAG_NO_MFMA false
#endif

namespace {

template <typename T> struct MfmaTile;

template <> struct MfmaTile<bf16_t> {
    // 16-byte chunk = 8 bf16 = one 16x16x32 operand fragment
    static __device__ __forceinline__ f32x4 mma(const uint4& w, const uint4& a, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w),
                                                       __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    }
};
template <> struct MfmaTile<float> {
    // 16-byte chunk = 4 fp32: lane (row, g) holds k = 4·chunk + e; the four 16x16x4 MFMAs each
    // contract the e-th element of every lane group (the k order inside a chunk is free as long
    // as both operands agree on it).
    static __device__ __forceinline__ f32x4 mma(const uint4& w, const uint4& a, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.x), __builtin_bit_cast(float, a.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.y), __builtin_bit_cast(float, a.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.z), __builtin_bit_cast(float, a.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.w), __builtin_bit_cast(float, a.w), c, 0, 0, 0);
        return c;
    }
};

template <bool FAST>
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case UIA_ACT_GELU: return gelu_erf_t<FAST>(x);
        case UIA_ACT_QUICKGELU: return quick_gelu(x);
        case UIA_ACT_RELU: return fmaxf(x, 0.0f);
        default: return x;
    }
}
// 8 values at a time: bf16-mode GELU goes through the packed polynomial, everything else through the scalar forms
template <bool FAST>
__device__ __forceinline__ void apply_act8(float (&v)[8], int act) {
    if (FAST && act == UIA_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 x = {v[e], v[e + 1]}, y = gelu_poly2(x);
            v[e] = y[0];
            v[e + 1] = y[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act<FAST>(v[e], act);
    }
}
template <bool FAST>
__device__ __forceinline__ float apply_dact(float pre, int act);
template <bool FAST>
__device__ __forceinline__ void apply_dact8(float (&v)[8], const float (&a)[8], int act) {
    if (FAST && act == UIA_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 x = {a[e], a[e + 1]}, d = dgelu_poly2(x);
            v[e] *= d[0];
            v[e + 1] *= d[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= apply_dact<FAST>(a[e], act);
    }
}
template <bool FAST>
__device__ __forceinline__ float apply_dact(float pre, int act) {
    switch (act) {
        case UIA_ACT_GELU: return dgelu_erf_t<FAST>(pre);
        case UIA_ACT_QUICKGELU: return dquick_gelu(pre);
        case UIA_ACT_RELU: return pre > 0.0f ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA issued through inline asm: hipcc models the builtin as an LDS store and drains it (s_waitcnt vmcnt(0)) in front
// of the next ds_read, which serialises the whole staging pipeline (seen in the .s of the first ping-pong build: the slot
// that issued the pieces took 2444 cycles against 600-730 for the others).  The asm form is invisible to that pass; the
// kernels that use it retire their pieces with their own counted s_waitcnt vmcnt(N) + barrier before any read.
// M0 carries the wave-uniform LDS byte address of the 1 KiB piece (lane i lands at +16·i); it is saved/restored around.
__device__ __forceinline__ void glds16_asm(const char* gsrc, char* lds_wave_base) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst_u)
                 : "memory");
}

// Row sums travel as 64-bit FIXED-POINT numbers (units of 2^-30): integer atomics add exactly, so (Σ, Σ²) do not depend on the order in
// which the column tiles of a row arrive — with float atomics they did, a 1e-7 wobble of a LayerNorm statistic re-rolled the bf16 roundings
// of everything downstream, and the step was not reproducible from run to run.  Range: Σ² up to 8.6e9 (row RMS 3300 at 768 columns).
typedef long long2 __attribute__((ext_vector_type(2)));
constexpr float ROWSUM_SCALE = 1073741824.0f, ROWSUM_INV = 1.0f / 1073741824.0f;
// A partial sum (one wave column of one launch) is kept below ROWSUM_PART_MAX: up to 16 of them per row cannot wrap the 64-bit sum, and
// llrintf never sees an infinity or a NaN (undefined).  A clamped partial means the row's statistics are garbage: bit 1 of the caller's
// guard word (uia_gemm_desc.ln_flag) says so, and the host fails loudly instead of training on a finite but wrong LayerNorm.
constexpr float ROWSUM_PART_MAX = 5.0e8f;
__device__ __forceinline__ void rowsum_add(void* dst, size_t row, float s1, float s2, int* flag) {
    if (!(fabsf(s1) < ROWSUM_PART_MAX) || !(s2 < ROWSUM_PART_MAX)) {
        if (flag) atomicOr(flag, 2);
        s1 = fminf(fmaxf(s1, -ROWSUM_PART_MAX), ROWSUM_PART_MAX);        // fmaxf(NaN, a) = a
        s2 = fminf(fmaxf(s2, 0.0f), ROWSUM_PART_MAX);
    }
    unsigned long long* q = (unsigned long long*)dst + 2 * row;
    atomicAdd(q, (unsigned long long)llrintf(s1 * ROWSUM_SCALE));        // two's complement: negative sums wrap correctly
    atomicAdd(q + 1, (unsigned long long)llrintf(s2 * ROWSUM_SCALE));
}
__device__ __forceinline__ float2 rowsum_load(const void* src, size_t row) {
    const long2 v = *(const long2*)((const long long*)src + 2 * row);
    return float2{(float)v[0] * ROWSUM_INV, (float)v[1] * ROWSUM_INV};
}
// (mean, rstd, -mean·rstd) of a row from its (Σ, Σ²) over `dim` columns, as the rowsum_out feature of a producing GEMM leaves them
struct LnRow { float mean, rstd, nmr; };
__device__ __forceinline__ LnRow ln_row_from_sums(float2 ss, int dim, float eps) {
    const float inv = 1.0f / (float)dim;
    const float mean = ss.x * inv;
    const float var = fmaxf(fmaf(-mean, mean, ss.y * inv), 0.0f);
    const float rstd = rsqrtf(var + eps);
    return LnRow{mean, rstd, -mean * rstd};
}
// Guard of a folded LayerNorm (uia_gemm_desc.ln_flag): the consumer's A operand was bf16(x), whose rounding error relative to the row's
// spread grows with |mean| / std.  Only an offending row issues the atomic.
__device__ __forceinline__ void ln_fold_guard(const UiaGemmParams& p, const LnRow& ln) {
    if (p.ln_flag && fabsf(ln.nmr) > (p.ln_flag_limit > 0.f ? p.ln_flag_limit : 8.0f)) atomicOr(p.ln_flag, 1);
}

// sum over the LPR consecutive lanes that share a row in the LDS-staged epilogue (LPR = 8: two quad permutes + a half-row mirror, all DPP)
template <int LPR>
__device__ __forceinline__ float row_lanes_sum(float v) {
    if constexpr (LPR == 8) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    } else {
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
    }
    return v;
}

template <typename T, int MT, int NT, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue(const UiaGemmParams& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wm, int wn, int li, int g) {
    // ---- epilogue: lane owns row m, columns nb .. nb+4NT-1 (nb multiple of 16)
    const int nb = n0 + wn * WTN + g * (4 * NT);
    T* outT = (T*)p.outT;
    const T* aux_in = (const T*)p.aux_in;
    T* aux_out = (T*)p.aux_out;
    const T* residT = (const T*)p.residT;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * WTM + 16 * i + li;
        if (m >= p.M) continue;
        const size_t orow = p.out_group > 0 ? (size_t)(m + m / p.out_group + 1) : (size_t)m;
        const size_t rrow = p.resid_mod > 0 ? (size_t)(m % p.resid_mod + p.resid_row_off) : orow;
#pragma unroll
        for (int jj = 0; jj < NT; jj += 2) {
            const int n = nb + 4 * jj;
            if (n >= p.N) continue;   // N is a multiple of 8 (checked on the host)
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[i][jj][e] * p.alpha; v[4 + e] = acc[i][jj + 1][e] * p.alpha; }
            if (p.lnfold_sums) {                          // LayerNorm folded into this GEMM: A held the raw rows (include/uia_hip.h)
                const float2 ss = rowsum_load(p.lnfold_sums, (size_t)m);
                const LnRow ln = ln_row_from_sums(ss, p.lnfold_dim, p.lnfold_eps);
                if (jj == 0 && nb == 0) ln_fold_guard(p, ln);
                float cs[8];
                load8(p.lnfold_colsum + n, cs);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], ln.rstd, ln.nmr * cs[e]);
            }
            if (p.bias) {
                float b[8];
                load8(p.bias + n, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += b[e];
            }
            if (aux_out) store8(aux_out + orow * p.ldaux_out + n, v);
            if (p.act) {
                apply_act8<sizeof(T) == 2>(v, p.act);
            }
            if (p.dact) {
                float a[8];
                load8(aux_in + orow * p.ldaux_in + n, a);
                apply_dact8<sizeof(T) == 2>(v, a, p.dact);
            }
            if (p.drop_where == 2) {
                const uint32_t keep = dropout_keep8(p.drop_seed, (uint32_t)(((size_t)m * (size_t)p.N + (size_t)n) >> 3), dropout_thresh16(p.drop_p));
                const float inv_keep = 1.0f / (1.0f - p.drop_p);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (keep >> e) & 1u ? v[e] * inv_keep : 0.f;
            }
            if (p.resid) {
                float r[8];
                load8(p.resid + rrow * p.ldr + n, r);
                if (p.resid_ln_stats) {                       // the residual is LayerNorm(resid row): same expression as ln_fwd_kernel
                    float2 ms;
                    if (p.resid_ln_dim > 0) { const LnRow ln = ln_row_from_sums(rowsum_load(p.resid_ln_stats, rrow), p.resid_ln_dim, p.resid_ln_eps); ms = float2{ln.mean, ln.rstd}; }
                    else ms = *(const float2*)((const float*)p.resid_ln_stats + 2 * rrow);
                    float lw[8], lb[8];
                    load8(p.resid_ln_w + n, lw);
                    load8(p.resid_ln_b + n, lb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = fmaf((r[e] - ms.x) * ms.y, lw[e], lb[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (residT) {
                float r[8];
                load8(residT + orow * p.ldrT + n, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            if (p.out32) store8(p.out32 + orow * p.ldo32 + n, v);
            if (outT) store8(outT + orow * p.ldo + n, v);
            if (p.rowsum_out) {                           // small-M configs: one pair of atomics per 8-column segment
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
                rowsum_add(p.rowsum_out, orow, s1, s2, p.ln_flag);
            }
        }
    }
}


// Epilogue of the ping-pong / ring kernels.  Every wave bounces its accumulators through a private LDS patch so that
// the bias/residual/aux loads and the C stores are issued with one lane per 8 consecutive columns and consecutive
// lanes on consecutive 16/32-byte pieces of a row: whole 128/256-byte row segments per wave instruction instead of
// 16 rows × 16-byte pieces in the MFMA layout (those stores were issue-bound: 17K-49K cycles per 256×256 tile).
//
// The patch holds 64 rows (4 MFMA row groups) at a time and the read-back/compute/store body is a ROLLED loop over
// those rows.  It used to be unrolled over all 16 passes of a tile with every runtime option inlined in each copy:
// ~18K instructions, >100 KB of straight-line code against a 64 KB instruction cache, and the stamps showed the
// epilogue taking 20K cycles per tile with its loads AND stores disabled (tests/test_gemm_stamps, diag 2) — it was
// instruction-fetch bound.  The K-loop's LDS buffers are dead by now, so the bigger patch costs nothing.
template <int MT, int WTN> struct EpiPatch {
    static constexpr int LDW = WTN + 4;                 // floats per staged row (+4 keeps the b128 accesses conflict-free)
    static constexpr int GPP = MT < 4 ? MT : (MT == 4 ? 2 : 4);   // MFMA row groups per phase (half-height tiles: 32-row patches, so that the
                                                                  // eight patches fit the 72 KB of a 3-deep ring and two workgroups share a CU)
    static constexpr int ROWS = GPP * 16;
    static constexpr int BYTES_PER_WAVE = ROWS * LDW * 4;
};

// EPI selects the epilogue's feature set at COMPILE time (bit mask of EPI_*; the launcher picks the instantiation that
// matches the descriptor) or, when EPI_GENERIC, at run time from the descriptor.  One training step uses six masks
// for >99% of its GEMM time (tools/gemm_census.py); with the features folded the body is branch-free and the
// compiler interleaves two read-back passes.  Specialised masks imply alpha == 1, no row remapping and GELU as the
// activation; anything else takes the generic instantiation.
enum : int { EPI_BIAS = 1, EPI_AUX_OUT = 2, EPI_GELU = 4, EPI_DGELU = 8, EPI_RESID = 16, EPI_RESIDT = 32, EPI_OUT32 = 64, EPI_OUTT = 128,
             EPI_RESID_LN = 256,          // with EPI_RESID: the residual is the LayerNorm of the rows of `resid` (resid_ln_stats / _w / _b)
             EPI_ROWSUM = 512,            // rowsum_out: (Σ, Σ²) of the stored fp32 rows, for the LayerNorm folded into the consuming GEMM
             EPI_LNFOLD = 1024,           // lnfold_*: A held raw rows, the LayerNorm is applied to the accumulators
             EPI_QUICK = 2048,            // with EPI_GELU / EPI_DGELU: the activation is QuickGELU (OpenAI CLIP towers: ViT-L/14 + LoRA, CLIPSeg), not GELU
             EPI_RESID_LO = 4096,         // the residual is a three-byte tensor: hi plane residT (bf16) + low bytes resid_lo8 (uia_gemm_desc.resid_lo8); resid_ln_* apply to it
             EPI_OUT_LO = 8192,           // the result is written as a three-byte tensor: hi plane outT + low bytes out_lo8
             EPI_GENERIC = -1 };

// WCOLS: wave columns of the workgroup that share a row block (their row-sum partials meet in LDS): 4 for the eight-wave kernels; the
// four-wave kernel (gemm_quad.hip) runs this epilogue once per 64-column half of its 128-wide wave tiles with WCOLS = 2.
template <typename T, int MT, int NT, int WTM, int WTN, int EPI = EPI_GENERIC, bool PATCH16 = false, int WCOLS = 4>
__device__ __forceinline__ void gemm_epilogue_lds(const UiaGemmParams& p, f32x4 (&acc)[MT][NT], char* smem, int wave, int lane, int m0, int n0,
                                                  int wm, int wn, float* lnrow_lds = nullptr, const float2* lnpre = nullptr) {
    constexpr bool GEN = EPI == EPI_GENERIC;
    const bool f_bias = GEN ? p.bias != nullptr : (EPI & EPI_BIAS) != 0;
    const bool f_aux_out = GEN ? p.aux_out != nullptr : (EPI & EPI_AUX_OUT) != 0;
    const bool f_resid = GEN ? p.resid != nullptr : (EPI & EPI_RESID) != 0;
    const bool f_rlo = GEN ? p.resid_lo8 != nullptr : (EPI & EPI_RESID_LO) != 0;          // three-byte residual: residT is its hi plane, not a second residual
    const bool f_olo = GEN ? p.out_lo8 != nullptr : (EPI & EPI_OUT_LO) != 0;
    const bool f_residT = GEN ? (p.residT != nullptr && p.resid_lo8 == nullptr) : ((EPI & EPI_RESIDT) != 0 && (EPI & EPI_RESID_LO) == 0);
    const bool f_rln = GEN ? ((p.resid != nullptr || p.resid_lo8 != nullptr) && p.resid_ln_stats != nullptr) : (EPI & EPI_RESID_LN) != 0;
    const bool f_out32 = GEN ? p.out32 != nullptr : (EPI & EPI_OUT32) != 0;
    const bool f_outT = GEN ? p.outT != nullptr : (EPI & EPI_OUTT) != 0;
    const bool f_rowsum = GEN ? p.rowsum_out != nullptr : (EPI & EPI_ROWSUM) != 0;
    const bool f_lnfold = GEN ? p.lnfold_sums != nullptr : (EPI & EPI_LNFOLD) != 0;
    const int act = GEN ? p.act : ((EPI & EPI_GELU) ? ((EPI & EPI_QUICK) ? UIA_ACT_QUICKGELU : UIA_ACT_GELU) : UIA_ACT_NONE);
    const int dact = GEN ? p.dact : ((EPI & EPI_DGELU) ? ((EPI & EPI_QUICK) ? UIA_ACT_QUICKGELU : UIA_ACT_GELU) : UIA_ACT_NONE);
    using EP = EpiPatch<MT, WTN>;
    constexpr int LDW = EP::LDW, GPP = EP::GPP, ROWS = EP::ROWS;
    constexpr int LPR = WTN / 8;                    // lanes per row when reading back
    constexpr int RPP = 64 / LPR;                   // rows per read pass
    static_assert(MT % GPP == 0, "row groups must split evenly into phases");
    const int li = lane & 15, g = lane >> 4;
    const int rr = lane / LPR, rc = (lane % LPR) * 8;
    const int n = n0 + wn * WTN + rc;
    T* outT = (T*)p.outT;
    const T* aux_in = (const T*)p.aux_in;
    T* aux_out = (T*)p.aux_out;
    const T* residT = (const T*)p.residT;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    if (f_bias && n < p.N) load8(p.bias + n, bias);
    const bool col_ok = n < p.N;
    float lnw[8], lnb[8];                           // deferred LayerNorm residual: the lane's eight columns of its weight and bias
#pragma unroll
    for (int e = 0; e < 8; ++e) { lnw[e] = 1.f; lnb[e] = 0.f; }
    if (f_rln && col_ok) { load8(p.resid_ln_w + n, lnw); load8(p.resid_ln_b + n, lnb); }
    float csum[8];                                  // folded LayerNorm: the lane's eight column sums of the pre-scaled weight
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    if (f_lnfold && col_ok) load8(p.lnfold_colsum + n, csum);
    // Folded LayerNorm: (rstd, -mean·rstd) of the wave's WTM rows, staged ONCE in the wave's own LDS strip.  Reading the row sums from
    // global memory inside the row loop put one L2 round trip on the critical path of every pass of an epilogue that has no other
    // load (QKV: 224 -> 247 us per launch); staged, the pass reads them like its accumulators.
    const int mrow0 = m0 + wm * WTM;
    // Deferred LayerNorm of the residual (EPI_RESID_LN): (mean, rstd) of the wave's rows staged the same way when the caller requested the
    // statistics ahead of its K loop (lnpre): the row passes then read them from LDS instead of waiting on one more load each.
    const bool rln_staged = f_rln && !f_lnfold && lnrow_lds != nullptr && lnpre != nullptr && !(GEN && (p.resid_mod > 0 || p.out_group > 0));
    if (rln_staged) {
#pragma unroll
        for (int i = 0; i < (WTM + 63) / 64; ++i) {
            const int r = lane + 64 * i;
            if (r < WTM) {
                float2 ms = lnpre[i];
                if (p.resid_ln_dim > 0) { const LnRow ln = ln_row_from_sums(ms, p.resid_ln_dim, p.resid_ln_eps); ms = float2{ln.mean, ln.rstd}; }
                *(float2*)(lnrow_lds + 2 * r) = ms;
            }
        }
    }
    if (f_lnfold && lnrow_lds) {
#pragma unroll
        for (int i = 0; i < (WTM + 63) / 64; ++i) {
            const int r = lane + 64 * i, m = mrow0 + r;
            if (r < WTM) {
                // lnpre: the caller requested the sums before its K loop (the ring kernel), so nothing waits on memory here
                const float2 ss = lnpre ? lnpre[i] : (m < p.M ? rowsum_load(p.lnfold_sums, (size_t)m) : float2{0.f, 1.f});
                const LnRow ln = ln_row_from_sums(ss, p.lnfold_dim, p.lnfold_eps);
                if (wn == 0 && n0 == 0 && m < p.M) ln_fold_guard(p, ln);        // once per row: the first column tile's first wave column
                *(float2*)(lnrow_lds + 2 * r) = float2{ln.rstd, ln.nmr};
            }
        }
    }

    // one row segment of 8 columns: bias / activation / residuals / stores.  pre_aux: the lane's eight aux_in values of this row, requested ahead
    // by the caller (bf16 dact epilogues: the load is otherwise one HBM round trip per pass on an epilogue that has nothing else to wait for)
    // pre_res / pre_rt: likewise the fp32 residual (two 16-byte halves) and the T residual.  os1 / os2: where the row sums of this pass go
    // instead of straight into their atomics (the caller issues those after the tile's last store: an atomic stays in the in-order
    // vmcnt queue for thousands of cycles under load, and every later load of the wave would wait behind it).
    auto apply = [&](const f32x4& lo, const f32x4& hi, int m, const bf16x8* pre_aux = nullptr, const f32x4* pre_res = nullptr,
                     const bf16x8* pre_rt = nullptr, float* os1 = nullptr, float* os2 = nullptr, const uint2* pre_lo = nullptr) {
        const bool ok = m < p.M && col_ok;
        if (!f_rowsum && !ok) return;
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const size_t orow = GEN && p.out_group > 0 ? (size_t)(m + m / p.out_group + 1) : (size_t)m;
        const size_t rrow = GEN && p.resid_mod > 0 ? (size_t)(m % p.resid_mod + p.resid_row_off) : orow;
        if (ok) {
        if (f_lnfold) {                                   // rstd·(x·W'ᵀ − mean·colsum): the LayerNorm of the raw rows that A held
            float2 st;                                    // (rstd, -mean·rstd)
            if (lnrow_lds) st = *(const float2*)(lnrow_lds + 2 * (m - mrow0));
            else { const LnRow ln = ln_row_from_sums(rowsum_load(p.lnfold_sums, (size_t)m), p.lnfold_dim, p.lnfold_eps); st = float2{ln.rstd, ln.nmr}; if (n == 0) ln_fold_guard(p, ln); }
            const f32x2 rs = {st.x, st.x}, nm = {st.y, st.y};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {              // bias folded in (alpha == 1 with lnfold): two packed fma per element pair
                const f32x2 c2 = {csum[e], csum[e + 1]}, b2 = {bias[e], bias[e + 1]}, a2 = {v[e], v[e + 1]};
                const f32x2 r2 = __builtin_elementwise_fma(a2, rs, __builtin_elementwise_fma(nm, c2, b2));
                v[e] = r2[0];
                v[e + 1] = r2[1];
            }
        } else if (GEN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], p.alpha, bias[e]);
        } else if (f_bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[e];
        }
        if (f_aux_out && UIA_EPI_STORES) store8(aux_out + orow * p.ldaux_out + n, v);
        if (act) apply_act8<sizeof(T) == 2>(v, act);
        if (dact && UIA_EPI_LOADS) {
            float a[8];
            if (pre_aux) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (float)(*pre_aux)[e];
            } else {
                load8(aux_in + orow * p.ldaux_in + n, a);
            }
            apply_dact8<sizeof(T) == 2>(v, a, dact);
        }
        if (GEN && p.drop_where == 2) {                   // dx += drop(s·q·A): the backward of the LoRA input dropout, same draw as the forward
            const uint32_t keep = dropout_keep8(p.drop_seed, (uint32_t)(((size_t)m * (size_t)p.N + (size_t)n) >> 3), dropout_thresh16(p.drop_p));
            const float inv_keep = 1.0f / (1.0f - p.drop_p);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (keep >> e) & 1u ? v[e] * inv_keep : 0.f;
        }
        if ((f_resid || f_rlo) && UIA_EPI_LOADS) {
            float r[8];
            if (f_rlo) {
                // three-byte residual: float bits = (hi_bits << 16) + (lo << 8), lo a signed byte (sign-magnitude floats are monotone as integers,
                // so the integer sum is the value 'lo' steps of 2^-8 ulp(bf16) away from hi, across an exponent boundary too)
                typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
                u16x8 hv;
                uint2 lv;
                if (pre_rt) { hv = __builtin_bit_cast(u16x8, *pre_rt); lv = *pre_lo; }
                else {
                    constexpr int G = 64 / (int)sizeof(T);
                    const T* hp = p.residT_kb_rows ? residT + ((size_t)(n / G) * (size_t)p.residT_kb_rows + rrow) * G + (n % G) : residT + rrow * p.ldrT + n;
                    hv = *(const u16x8*)hp;
                    lv = *(const uint2*)(p.resid_lo_kb_rows ? p.resid_lo8 + (((size_t)(n >> 6) * (size_t)p.resid_lo_kb_rows + rrow) << 6) + (n & 63) : p.resid_lo8 + rrow * p.ld_resid_lo + n);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int lb = __builtin_amdgcn_sbfe((int)(e < 4 ? lv.x : lv.y), 8 * (e & 3), 8);
                    r[e] = __builtin_bit_cast(float, ((unsigned)hv[e] << 16) + ((unsigned)lb << 8));
                }
            } else if (pre_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = pre_res[0][e]; r[4 + e] = pre_res[1][e]; }
            } else {
                load8(p.resid + rrow * p.ldr + n, r);
            }
            if (f_rln) {                                  // same expression, same operands as ln_fwd_kernel: bit-identical to reading its fp32 output
                float2 ms;
                if (rln_staged) ms = *(const float2*)(lnrow_lds + 2 * (m - mrow0));
                else if (p.resid_ln_dim > 0) { const LnRow ln = ln_row_from_sums(rowsum_load(p.resid_ln_stats, rrow), p.resid_ln_dim, p.resid_ln_eps); ms = float2{ln.mean, ln.rstd}; }
                else ms = *(const float2*)((const float*)p.resid_ln_stats + 2 * rrow);
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = fmaf((r[e] - ms.x) * ms.y, lnw[e], lnb[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (f_residT && UIA_EPI_LOADS) {
            float r[8];
            if (pre_rt) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (float)(*pre_rt)[e];
            } else {
                load8(residT + orow * p.ldrT + n, r);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (f_out32 && UIA_EPI_STORES) store8(p.out32 + orow * p.ldo32 + n, v);
        if (f_olo && UIA_EPI_STORES) {                    // low bytes of the three-byte result: ((bits + 0x80) >> 8) - (bf16 bits << 8), clamped to a signed byte
            unsigned w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned vb = __builtin_bit_cast(unsigned, v[e]);
                const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)v[e]);
                int d = (int)((vb + 0x80u) >> 8) - (int)(hb << 8);
                d = d < -127 ? -127 : (d > 127 ? 127 : d);
                w[e] = (unsigned)d & 0xFFu;
            }
            uint2 pk;
            pk.x = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
            pk.y = w[4] | (w[5] << 8) | (w[6] << 16) | (w[7] << 24);
            *(uint2*)(p.out_lo_kb_rows ? p.out_lo8 + (((size_t)(n >> 6) * (size_t)p.out_lo_kb_rows + orow) << 6) + (n & 63) : p.out_lo8 + orow * p.ld_out_lo + n) = pk;
        }
        if (f_outT && UIA_EPI_STORES) {
            if (p.outT_kb_rows) {                         // K-blocked for the GEMM that reads it as A: 64-byte column blocks, rows contiguous inside a block
                constexpr int G = 64 / (int)sizeof(T);
                store8(outT + ((size_t)(n / G) * (size_t)p.outT_kb_rows + orow) * G + (n % G), v);
            } else {
                store8(outT + orow * p.ldo + n, v);
            }
        }
        }
        if (f_rowsum) {                                   // every lane takes part (DPP), rows / columns past the edge contribute zero
            float s1 = 0.f, s2 = 0.f;
            if (ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
            }
            s1 = row_lanes_sum<LPR>(s1);
            s2 = row_lanes_sum<LPR>(s2);
            if (os1) { *os1 = s1; *os2 = s2; }
            else if ((lane % LPR) == 0 && m < p.M) rowsum_add(p.rowsum_out, orow, s1, s2, p.ln_flag);
        }
    };

    if constexpr (PATCH16) {
        // Persistent kernel: the next tile's first sub-tiles are already landing in ring buffers 0..2, so the patch is
        // 16 rows × WTN floats per wave (8 waves = one 32 KiB ring buffer, the caller passes its base), un-padded, with the
        // 16-byte chunk c of row r stored at chunk c ^ (r & 15): conflict-free for the b128 writes (8 rows × one chunk
        // column) and for both b128 reads of a pass (rows r, r+1 take disjoint chunk sets).
        static_assert(WTN == 64 && NT == 4, "swizzled 16-row patch is laid out for 64-column wave tiles");
        float* stg = (float*)(smem + wave * (16 * WTN * 4));
        const int c0 = (lane % LPR) * 2;
#pragma clang loop unroll(full)
        for (int i = 0; i < MT; ++i) {
#pragma clang loop unroll(full)
            for (int j = 0; j < NT; ++j) *(f32x4*)(stg + li * WTN + (((g * NT + j) ^ li) << 2)) = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 16 / RPP; ++ps) {
                const int row = ps * RPP + rr;
                const f32x4 lo = *(const f32x4*)(stg + row * WTN + ((c0 ^ row) << 2));
                const f32x4 hi = *(const f32x4*)(stg + row * WTN + (((c0 + 1) ^ row) << 2));
                apply(lo, hi, m0 + wm * WTM + 16 * i + row);
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        float* stg = (float*)(smem + wave * EP::BYTES_PER_WAVE);
        // PIPELINED row passes (bf16, compile-time masks with an operand to read or row sums to leave).  The epilogue's loads, stores and
        // atomics share the wave's in-order vmcnt queue: a pass that loads its aux_in / residual rows in line waits for the PREVIOUS pass's
        // stores (and atomics) to be acknowledged first — one HBM round trip per pass, 16 per tile (in-kernel stamps, round 3: the GELU'
        // epilogue of the fc2 data gradient took 36 K cycles, 5.8 K of them arithmetic; the fold producer's 71 K).  Here the operand rows are
        // requested a CHUNK (half a 64-row phase) at a time into two register sets, two chunks ahead of their use, and the row-sum atomics are
        // issued after the tile's last store.  Same arithmetic in the same order: results are bit-identical to the in-line form.
#ifdef UIA_NO_PREAUX
        constexpr bool PIPE = false;                       // A/B build (tests/test_gemm_stamps_nopre)
#else
        constexpr bool PIPE = !GEN && sizeof(T) == 2 && (EPI & (EPI_DGELU | EPI_RESID | EPI_RESIDT | EPI_ROWSUM | EPI_RESID_LO)) != 0;
#endif
        constexpr int NPASS = ROWS / RPP, NPH = MT / GPP;
        // PACKED bounce (bf16, compile-time masks whose only outputs are T tensors and that read nothing: data gradients, QKV, fc1): the lane
        // applies the epilogue's arithmetic to its 16 consecutive columns of a row IN THE MFMA LAYOUT — the per-column vectors are fixed per
        // lane for the whole tile — and bounces the ROUNDED values through LDS (128 B per row instead of 256): half the LDS bytes of the fp32
        // bounce (ds_write_b128 moves ≈ 79 B/clk per CU: 3.3 K of a store-only epilogue's ≈ 8 K cycles), and the read-back passes have nothing
        // left to do but store.  Same arithmetic on the same values in the same order as the row passes: bit-identical results.
#if defined(UIA_NO_PREAUX) || defined(UIA_NO_PACK)
        constexpr bool PACK = false;                       // A/B builds
#else
        constexpr bool PACK = !GEN && sizeof(T) == 2 && NT == 4 && WTN == 64 && (EPI & EPI_OUTT) != 0 &&
                              (EPI & (EPI_RESID | EPI_RESIDT | EPI_OUT32 | EPI_DGELU | EPI_ROWSUM | EPI_RESID_LN | EPI_AUX_OUT)) == 0;
        // (masks with an aux_out stash stay on the fp32 bounce: two packed bounces per phase measured 275 vs 266 us on fc1's launch, same box)
#endif
        if constexpr (PACK) {
            constexpr int LDB = 128 + 16;                              // bytes per staged row of 64 bf16 (+16: the b128 writes of 8 consecutive rows hit 32 distinct banks)
            static_assert(ROWS * LDB <= EP::BYTES_PER_WAVE, "bf16 patch must fit the wave's fp32 patch");
            char* stb = (char*)stg;
            const int nl = n0 + wn * WTN + g * 16;                     // the lane's 16 columns in the MFMA layout
            const bool lcol_ok = nl < p.N;                             // N is a multiple of 8: the second half is tested on its own
            const bool lcol_ok2 = nl + 8 < p.N;
            float b16[16], c16[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { b16[e] = 0.f; c16[e] = 0.f; }
            if (f_bias) {
                if (lcol_ok) { float t8[8]; load8(p.bias + nl, t8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) b16[e] = t8[e]; }
                if (lcol_ok2) { float t8[8]; load8(p.bias + nl + 8, t8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) b16[8 + e] = t8[e]; }
            }
            if (f_lnfold) {
                if (lcol_ok) { float t8[8]; load8(p.lnfold_colsum + nl, t8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) c16[e] = t8[e]; }
                if (lcol_ok2) { float t8[8]; load8(p.lnfold_colsum + nl + 8, t8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) c16[8 + e] = t8[e]; }
            }
            // one bounce: `which` = 0 the T output (after the activation), 1 the aux_out stash (before it)
            auto bounce = [&](int ph, int which) {
#pragma clang loop unroll(full)
                for (int gi = 0; gi < GPP; ++gi) {
                    const int i = ph * GPP + gi;
                    const int mloc = 16 * i + li;                      // row within the wave's WTM rows
                    float2 st = float2{1.f, 0.f};
                    if (f_lnfold) st = *(const float2*)(lnrow_lds + 2 * mloc);
                    const f32x2 rs2 = {st.x, st.x}, nm2 = {st.y, st.y};
                    bf16x8 outv[2];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float v[8] = {acc[i][2 * hf][0], acc[i][2 * hf][1], acc[i][2 * hf][2], acc[i][2 * hf][3],
                                      acc[i][2 * hf + 1][0], acc[i][2 * hf + 1][1], acc[i][2 * hf + 1][2], acc[i][2 * hf + 1][3]};
                        if (f_lnfold) {
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const f32x2 c2 = {c16[8 * hf + e], c16[8 * hf + e + 1]}, bb2 = {b16[8 * hf + e], b16[8 * hf + e + 1]}, a2 = {v[e], v[e + 1]};
                                const f32x2 r2 = __builtin_elementwise_fma(a2, rs2, __builtin_elementwise_fma(nm2, c2, bb2));
                                v[e] = r2[0];
                                v[e + 1] = r2[1];
                            }
                        } else if (f_bias) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += b16[8 * hf + e];
                        }
                        if (which == 0 && act) apply_act8<true>(v, act);
                        const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
                        bf16x8 r = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3], (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
                        outv[hf] = r;
                    }
#ifdef UIA_EPI_DIRECT
                    // experiment build (tools/attic/ab_direct.sh): the rounded values go to memory from the MFMA layout (a lane holds 16 consecutive columns of one row: two
                    // 16-byte stores), no bounce.  Measured SLOWER, round 5: 65 536 x 2304 x 768 1052 -> 998 TF/s on cfg 8, 1083 -> 1029 on cfg 27, the step 40.1 -> 40.5-40.7 ms
                    // (profiles/r05_f_quadv_ablation.txt): a store instruction of sixteen rows x four 16-byte pieces costs the CU's store path more than the bounce's LDS time.
                    {
                        T* dbase = which == 0 ? outT : aux_out;
                        const int m = m0 + wm * WTM + mloc;
                        if (m < p.M && UIA_EPI_STORES) {
                            T* d = (which == 0 && p.outT_kb_rows) ? dbase + ((size_t)(nl / 32) * (size_t)p.outT_kb_rows + (size_t)m) * 32 + (nl % 32)
                                                                  : dbase + (size_t)m * (which == 0 ? p.ldo : p.ldaux_out) + nl;
                            if (lcol_ok) *(bf16x8*)d = outv[0];
                            if (lcol_ok2) *(bf16x8*)(d + 8) = outv[1];
                        }
                    }
                    continue;
#endif
                    *(bf16x8*)(stb + (gi * 16 + li) * LDB + g * 32) = outv[0];
                    *(bf16x8*)(stb + (gi * 16 + li) * LDB + g * 32 + 16) = outv[1];
                }
#ifdef UIA_EPI_DIRECT
                return;
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                T* dstbase = which == 0 ? outT : aux_out;
                const int mbase = m0 + wm * WTM + ph * ROWS + rr;
#pragma unroll
                for (int q = 0; q < NPASS; ++q) {
                    const int row = q * RPP + rr;
                    const int m = mbase + q * RPP;
                    const bf16x8 val = *(const bf16x8*)(stb + row * LDB + rc * 2);
                    if (m < p.M && col_ok && UIA_EPI_STORES) {
                        if (which == 0 && p.outT_kb_rows) {
                            constexpr int G = 32;
                            *(bf16x8*)(dstbase + ((size_t)(n / G) * (size_t)p.outT_kb_rows + (size_t)m) * G + (n % G)) = val;
                        } else {
                            *(bf16x8*)(dstbase + (size_t)m * (which == 0 ? p.ldo : p.ldaux_out) + n) = val;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            };
#pragma clang loop unroll(full)
            for (int ph = 0; ph < NPH; ++ph) {
                if (f_aux_out) bounce(ph, 1);
                bounce(ph, 0);
            }
        } else
        if constexpr (PIPE && NPASS % 2 == 0) {
            constexpr bool H_AUX = (EPI & EPI_DGELU) != 0, H_RES = (EPI & EPI_RESID) != 0, H_LO = (EPI & EPI_RESID_LO) != 0, H_RT = (EPI & EPI_RESIDT) != 0 || H_LO,
                           H_SUM = (EPI & EPI_ROWSUM) != 0;
            constexpr int CH = NPASS / 2, NCH = 2 * NPH;                 // passes per chunk, chunks per tile
            bf16x8 pa[2][H_AUX ? CH : 1];
            f32x4 pr[2][H_RES ? 2 * CH : 1];
            bf16x8 prt[2][H_RT ? CH : 1];
            uint2 plo[2][H_LO ? CH : 1];
            float rs1[H_SUM ? NCH * CH : 1], rs2[H_SUM ? NCH * CH : 1];
            const int mrow = m0 + wm * WTM + rr;
            const int nc = col_ok ? n : 0;
            auto issue = [&](int c, int set) {                          // chunk c = passes [c·CH, (c+1)·CH) of the wave's WTM / RPP row passes
                if (!UIA_EPI_LOADS) return;
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int m = mrow + (c * CH + q) * RPP;
                    const size_t mc = (size_t)(m < p.M ? m : p.M - 1);
                    if constexpr (H_AUX) pa[set][q] = *(const bf16x8*)((const bf16_t*)p.aux_in + mc * p.ldaux_in + nc);
                    if constexpr (H_RES) { const float* src = p.resid + mc * p.ldr + nc; pr[set][2 * q] = *(const f32x4*)src; pr[set][2 * q + 1] = *(const f32x4*)(src + 4); }
                    if constexpr (H_LO) {
                        const bf16_t* hp = p.residT_kb_rows ? (const bf16_t*)p.residT + ((size_t)(nc / 32) * (size_t)p.residT_kb_rows + mc) * 32 + (nc % 32)
                                                            : (const bf16_t*)p.residT + mc * p.ldrT + nc;
                        prt[set][q] = *(const bf16x8*)hp;
                        plo[set][q] = *(const uint2*)(p.resid_lo_kb_rows ? p.resid_lo8 + (((size_t)(nc >> 6) * (size_t)p.resid_lo_kb_rows + mc) << 6) + (nc & 63) : p.resid_lo8 + mc * p.ld_resid_lo + nc);
                    } else if constexpr (H_RT) prt[set][q] = *(const bf16x8*)((const bf16_t*)p.residT + mc * p.ldrT + nc);
                }
            };
            issue(0, 0);
            issue(1, 1);
#pragma clang loop unroll(full)
            for (int c = 0; c < NCH; ++c) {
                const int ph = c / 2, set = c & 1;
                if ((c & 1) == 0) {
#pragma clang loop unroll(full)
                    for (int gi = 0; gi < GPP; ++gi)
#pragma clang loop unroll(full)
                        for (int j = 0; j < NT; ++j) *(f32x4*)(stg + (gi * 16 + li) * LDW + g * (4 * NT) + 4 * j) = acc[ph * GPP + gi][j];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int row = ((c & 1) * CH + q) * RPP + rr;
                    const f32x4 lo = *(const f32x4*)(stg + row * LDW + rc), hi = *(const f32x4*)(stg + row * LDW + rc + 4);
                    apply(lo, hi, mrow + (c * CH + q) * RPP, H_AUX ? &pa[set][q] : nullptr, H_RES ? &pr[set][2 * q] : nullptr, H_RT ? &prt[set][q] : nullptr,
                          H_SUM ? &rs1[c * CH + q] : nullptr, H_SUM ? &rs2[c * CH + q] : nullptr, H_LO ? &plo[set][q] : nullptr);
                }
                if (c + 2 < NCH) issue(c + 2, set);
                if ((c & 1) == 1) __builtin_amdgcn_wave_barrier();
            }
            if constexpr (H_SUM) {
                // Row sums of the tile: the four waves that share a row block (wn = 0..3) meet in LDS, then ONE 64-lane atomic instruction per wave adds
                // the tile's 2 x (rows of the workgroup) sums — the first form issued two 8-lane atomic instructions per row pass and wave column
                // (256 instructions per 256 x 256 tile, 4 x the atomics; ~15 us of a 176 us launch).  Partials meet in a fixed order: still bit-reproducible.
                static_assert(WTN == 64, "64-column wave tiles");
                constexpr int RW = 2 * WTM;                                   // values a wave leaves: (Σ, Σ²) of its WTM rows
                // each partial goes to fixed point on its own (exactly what its atomic would have added), the integers are summed: bit-identical to
                // the in-line form, whatever the order
                auto to_fixed = [&](float v, bool sq) -> long long {
                    if (!(fabsf(v) < ROWSUM_PART_MAX)) {
                        if (p.ln_flag) atomicOr(p.ln_flag, 2);
                        v = fminf(fmaxf(v, sq ? 0.0f : -ROWSUM_PART_MAX), ROWSUM_PART_MAX);
                    }
                    return llrintf(v * ROWSUM_SCALE);
                };
                long long* ex = (long long*)smem;                             // the patches are dead: [waves][RW]
                __syncthreads();                                              // every wave is done with its patch
                if ((lane % LPR) == 0) {
#pragma unroll
                    for (int k = 0; k < NCH * CH; ++k) {
                        long long* d = ex + wave * RW + 2 * (k * RPP + rr);
                        d[0] = to_fixed(rs1[k], false);
                        d[1] = to_fixed(rs2[k], true);
                    }
                }
                __syncthreads();
                const int nwaves = (int)(blockDim.x >> 6), wm_n = nwaves / WCOLS;  // wave rows of the workgroup
                for (int idx = wave * 64 + lane; idx < wm_n * RW; idx += nwaves * 64) {
                    const int wmr = idx / RW, off = idx - wmr * RW;           // off = 2 * row + which
                    const long long* src = ex + (wmr * WCOLS) * RW + off;
                    long long v = src[0];
#pragma unroll
                    for (int c = 1; c < WCOLS; ++c) v += src[c * RW];         // integers: any order gives the same sum
                    const int m = m0 + wmr * WTM + (off >> 1);
                    if (m < p.M) atomicAdd((unsigned long long*)p.rowsum_out + 2 * (size_t)m + (off & 1), (unsigned long long)v);
                }
            }
        } else {
#pragma clang loop unroll(full)
        for (int ph = 0; ph < MT / GPP; ++ph) {
#pragma clang loop unroll(full)
            for (int gi = 0; gi < GPP; ++gi)
#pragma clang loop unroll(full)
                for (int j = 0; j < NT; ++j) *(f32x4*)(stg + (gi * 16 + li) * LDW + g * (4 * NT) + 4 * j) = acc[ph * GPP + gi][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int mbase = m0 + wm * WTM + ph * ROWS + rr;
#pragma clang loop unroll_count(2)
            for (int q = 0; q < ROWS / RPP; ++q) {
                const int row = q * RPP + rr;
                const f32x4 lo = *(const f32x4*)(stg + row * LDW + rc), hi = *(const f32x4*)(stg + row * LDW + rc + 4);
                apply(lo, hi, mbase + q * RPP);
            }
            __builtin_amdgcn_wave_barrier();
        }
        }
    }
}

// feature mask of a descriptor, or EPI_GENERIC when it uses something the specialised epilogues leave out
inline int epi_mask_of(const UiaGemmParams& p) {
    if (p.alpha != 1.0f || p.out_group > 0 || p.resid_mod > 0 || p.drop_where == 2) return EPI_GENERIC;
    const bool quick = p.act == UIA_ACT_QUICKGELU || p.dact == UIA_ACT_QUICKGELU;
    if ((p.act && p.act != UIA_ACT_GELU && p.act != UIA_ACT_QUICKGELU) || (p.dact && p.dact != UIA_ACT_GELU && p.dact != UIA_ACT_QUICKGELU) ||
        (p.act && p.dact && p.act != p.dact)) return EPI_GENERIC;
    return (quick ? EPI_QUICK : 0) | (p.bias ? EPI_BIAS : 0) | (p.aux_out ? EPI_AUX_OUT : 0) | (p.act ? EPI_GELU : 0) | (p.dact ? EPI_DGELU : 0) | (p.resid ? EPI_RESID : 0) |
           ((p.resid && p.resid_ln_stats) ? EPI_RESID_LN : 0) | (p.rowsum_out ? EPI_ROWSUM : 0) | (p.lnfold_sums ? EPI_LNFOLD : 0) |
           ((p.residT && !p.resid_lo8) ? EPI_RESIDT : 0) | (p.out32 ? EPI_OUT32 : 0) | (p.outT ? EPI_OUTT : 0) |
           (p.resid_lo8 ? EPI_RESID_LO : 0) | ((p.resid_lo8 && p.resid_ln_stats) ? EPI_RESID_LN : 0) | (p.out_lo8 ? EPI_OUT_LO : 0);
}

}  // namespace
