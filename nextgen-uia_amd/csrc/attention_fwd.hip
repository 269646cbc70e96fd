// attention_fwd.hip — softmax(q kᵀ·scale + mask) v for short sequences (L ≤ 272), one
// workgroup per (batch, head); the whole K and V of a head live in LDS (single pass, no
// online-softmax tiling).
//
// Replaces: timm Attention / F.scaled_dot_product_attention (BiomedCLIP vision tower),
//           nn.MultiheadAttention in /root/reference/src/third_party/openai_clip/model.py:195-197,
//           the SDPA call of /root/reference/src/adapters/lora.py:188,
//           HF BertSelfAttention (key-padding mask) and the causal CLIP text blocks (model.py:346-352).
//
// Layout: q/k/v are read in place from the projection output: element (b, l, h, d) lives at
// ptr[(b*L + l)*ld + h*64 + d]  (fused qkv → three base pointers 768 columns apart).  The output
// uses the same convention with its own leading dimension.  Head dim is fixed at 64.
//
// bf16 path (MFMA, wave64):
//   Sᵀ = K·Qᵀ is computed "swapped" so that each lane owns ONE query column and 4·LT of its keys:
//   the row max / row sum are in-lane reductions plus two xor-shuffles (16, 32).  The exponentiated
//   accumulator tiles are then, without any lane movement, the B operand of Oᵀ = Vᵀ·Pᵀ (k-slot
//   (g,e) of a 32-key block ↔ key 32u + 16(e>>2) + 4g + (e&3)); the matching Vᵀ A-operand is
//   fetched from the row-major V tile with ds_read_b64_tr_b16 (hardware transpose), so V is never
//   transposed in memory.
// fp32 path (parity mode): plain VALU two-pass softmax, one query per thread.
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include "uia_common.h"
#include "uia_kernels.h"

#ifdef AFWD_STAMPS
// diagnostic build (tools/afwd_stamps.sh): cycles per phase of the bf16 forward for the waves of ONE workgroup:
// [0] issue of the Q / K / V requests, [1] wait + barrier, [2] S = K·Qᵀ + row max, [3] exp / row sum, [4] Oᵀ = Vᵀ·Pᵀ, [5] stores, [6] kernel
__device__ unsigned long long uia_afwd_stamps[8 * 8];
extern "C" int uia_afwd_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(uia_afwd_stamps), sizeof(uia_afwd_stamps));
}
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
// the same through an LDS-space pointer: the fragment address is then the lane's base register + an immediate offset, not a generic pointer rebuilt per read
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ s16x4 lds_tr16(const lds_char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p); }

// ------------------------------------------------------------------------------------------
// LT_MAX = max number of 16-key tiles (L ≤ 16·LT_MAX).  NP = 32-key blocks.
// NW = waves per workgroup: the query tiles of a head are dealt round-robin to the waves.  Four waves for the 13 tiles of a 197-token
// head meant 4 + 3 + 3 + 3 tiles after each other per workgroup; seven waves take two tiles each (one wave a single one), eight
// waves the 16 tiles of a 256-token caption: the workgroup's critical path is staging + 2 tiles (ViT-B layer 126 -> 108 us, BERT
// 93 -> 74 us).  Thirteen waves with one tile each leave one workgroup per CU and are slower (124 us); waiting for K and V
// separately (scores under the V landing, LDS-DMA from inline asm with counted vmcnt) was slower too (120 us).
template <int LT_MAX, int NW>
__global__ __launch_bounds__(64 * NW, NW >= 7 ? 4 : 1) void attn_fwd_bf16_kernel(const UiaAttnParams p) {   // two 7/8-wave workgroups per CU: ≤ 128 VGPRs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP_MAX = (LT_MAX + 1) / 2;
    const int bb = blockIdx.x / p.H;
    // packed (un-padded) sequences: this head's rows start at cu_seqlens[b] and there are cu[b+1]-cu[b] of them, all valid
    const int L = p.cu_seqlens ? p.cu_seqlens[bb + 1] - p.cu_seqlens[bb] : p.L;
    const int LT = (L + 15) >> 4;           // key / query tiles in use
    char* Ks = smem;                        // [16·4·ceil(LT/4)][128 B], 16-B chunk swizzle (row>>1)&7: whole chunks of four key tiles, so that a chunk that runs past
                                            // the last tile still reads inside the allocation and every fragment address is the lane's base + a compile-time offset
    char* Vs = smem + ((LT + 3) >> 2) * 4 * 2048;   // [32·ceil(LT/2)][128 B], 16-B chunk swizzle ((row>>1)&3)<<1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AFWD_STAMPS
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif
    const int b = bb, h = blockIdx.x - b * p.H;
    const size_t row0 = p.cu_seqlens ? (size_t)p.cu_seqlens[b] : (size_t)b * L;
    const char* qb = (const char*)p.q + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* kb = (const char*)p.k + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* vb = (const char*)p.v + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const size_t rs = (size_t)p.ld_qkv * 2;  // row stride in bytes

    // Keys past the last valid one contribute exactly 0: with a key-padding mask only the 16-key tiles that hold a valid
    // key are staged, multiplied and exponentiated (BERT captions fill 24-128 of 256 positions); a causal mask stops
    // each query tile at its own diagonal tile.
    int klen = L;
    if (p.mask_kind == UIA_MASK_KEYPAD && p.keylen) { klen = p.keylen[b]; klen = klen < 1 ? 1 : (klen > L ? L : klen); }
    const int LTk = (klen + 15) >> 4;        // key tiles in use (wave-uniform: depends on the batch row only)
    const int NPk = (LTk + 1) >> 1;

    // ---- this wave's query fragments (tiles wave, wave+4, ...), fetched straight from HBM BEFORE the K/V wait: their latency
    //      (1-2 us per tile when loaded at the top of each tile's iteration, a third of the workgroup's life) hides under the staging.
    constexpr int NQT = (LT_MAX + NW - 1) / NW;
    const int li_q = lane & 15, g_q = lane >> 4;
    uint4 qa[NQT], qb2[NQT];
#pragma unroll
    for (int i = 0; i < NQT; ++i) {
        const int qt = wave + NW * i;
        int qr = 16 * qt + li_q;
        qr = qr < L ? qr : L - 1;
        qa[i] = *(const uint4*)(qb + qr * rs + g_q * 16);
        qb2[i] = *(const uint4*)(qb + qr * rs + (g_q + 4) * 16);
    }
    // ---- stage K and V (rows past L are clamped to row L-1: finite, and masked / multiplied by 0)
    const int ninstr = (NPk * 32) >> 3;      // 1 KiB wave-instructions per tensor
    for (int q = wave; q < ninstr; q += NW) {
        const int r = 8 * q + (lane >> 3);
        const int gr = r < L ? r : L - 1;
        const int ck = (lane & 7) ^ ((r >> 1) & 7);
        const int cv = (lane & 7) ^ (((r >> 1) & 3) << 1);
        glds16(kb + gr * rs + ck * 16, Ks + q * 1024);
        glds16(vb + gr * rs + cv * 16, Vs + q * 1024);
    }
#ifdef AFWD_STAMPS
    const unsigned long long st_issued = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef AFWD_STAMPS
    const unsigned long long st_staged = __builtin_amdgcn_s_memtime();
#endif

    const int li = lane & 15, g = lane >> 4;
    const float sc = p.scale * 1.44269504088896341f;   // softmax in base 2

    // K fragment (MFMA A operand, rows = keys): row 16t+li, chunk g+4kk
    const int offK0 = li * 128 + ((g ^ (li >> 1)) << 4);
    // Vᵀ fragment via transpose read: group g reads rows 32u + 16hh + 4g + q', cols 16dt + 4pp
    const int qq = li >> 2, pp = li & 3;
    const int vrow0 = 4 * g + qq;                       // + 32u + 16hh  (multiples of 16 keep (row>>1)&3)
    const int vsw = (vrow0 >> 1) & 3;
    const lds_char* vfrag[4];                          // column group dt of row vrow0: + 32·128·u (+ 16·128) per fragment, compile-time
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vfrag[dt] = (const lds_char*)Vs + vrow0 * 128 + ((dt ^ vsw) << 5) + 8 * pp;

#pragma unroll
    for (int qi = 0; qi < NQT; ++qi) {
        const int qt = wave + NW * qi;
        if (qt >= LT) break;
        const int qrow = 16 * qt + li;
        const uint4 q0 = qa[qi], q1 = qb2[qi];
#ifdef AFWD_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif

        const int LTq = (p.mask_kind == UIA_MASK_CAUSAL && qt + 1 < LTk) ? qt + 1 : LTk;   // key tiles this query tile can see
        const int NPq = (LTq + 1) >> 1;
        // Tiles are skipped in chunks of 4 (one wave-uniform branch per chunk keeps 4 tiles of independent work in a
        // basic block; a branch per tile cost more than the skipped work).  A chunk may run past LTq: its extra tiles
        // read K rows that were never staged, and the select below discards whatever they produce.
        constexpr int NCH = (LT_MAX + 3) / 4;
        const int kmax = p.mask_kind == UIA_MASK_CAUSAL ? (qrow < klen - 1 ? qrow : klen - 1) : klen - 1;  // last valid key
        f32x4 s[4 * NCH];
        float m = -INFINITY;                                     // row max of the RAW scores (scale > 0: the scaled max is m·sc)
        // one key tile: scores of 16 keys for this lane's query.  TEST = false for chunks that lie entirely inside the valid keys of a
        // non-causal head: no per-element compare / select (the 197-token image heads spend 12 of their 13 tiles there).
        auto tile = [&](int t, auto test) {
            const uint4 k0 = *(const uint4*)(Ks + t * 2048 + offK0);          // tiles past the staged ones hold garbage: their scores are discarded by the select
            const uint4 k1 = *(const uint4*)(Ks + t * 2048 + (offK0 ^ 64));
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k0), __builtin_bit_cast(bf16x8, q0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k1), __builtin_bit_cast(bf16x8, q1), acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {                             // lane owns query qrow, keys 16t + 4g + r
                if (decltype(test)::value) acc[r] = (16 * t + 4 * g + r) <= kmax ? acc[r] : -INFINITY;
                m = fmaxf(m, acc[r]);
            }
            s[t] = acc;
        };
        const int interior = p.mask_kind == UIA_MASK_CAUSAL ? 0 : klen >> 6;     // chunks of 4 tiles (64 keys) with no masked key
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (4 * c < LTq) {
                // (tiles at and past LT_MAX cannot hold a key of this instantiation — L <= 16·LT_MAX by dispatch: the last chunk of the 13-tile kernel is ONE tile,
                //  not four of which three were computed, exponentiated and discarded: 13 tiles instead of 16 per query tile of a 197-token head, round 6)
                if (c < interior) {
#pragma unroll
                    for (int t = 4 * c; t < 4 * c + 4; ++t) if (t < LT_MAX) tile(t, std::false_type{});
                } else {
#pragma unroll
                    for (int t = 4 * c; t < 4 * c + 4; ++t) if (t < LT_MAX) tile(t, std::true_type{});
                }
            } else {
#pragma unroll
                for (int t = 4 * c; t < 4 * c + 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        m = rows_max(m);                                         // the four lanes that share this query (permlane swaps: no LDS trip)
#ifdef AFWD_STAMPS
        asm volatile("" :: "v"(m));
        const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
        const float msc = m * sc;                                // softmax in base 2: exp2(s·sc − m·sc), one fma per element
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (4 * c < LTq) {
#pragma unroll
                for (int t = 4 * c; t < 4 * c + 4; ++t)
                    if (t < LT_MAX) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], sc, -msc));      // exp2(-inf) = 0 for masked keys
                            s[t][r] = e;
                            sum += e;
                        }
                    } else {
                        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};         // never a key: contributes 0 to the PV product's padded k-slots
                    }
            }
        }
        sum = rows_sum(sum);
        const float inv = 1.0f / sum;
#ifdef AFWD_STAMPS
        asm volatile("" :: "v"(inv));
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif

        // ---- Oᵀ = Vᵀ · Pᵀ
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NP_MAX; ++u) {
            if (u < NPq) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[e] = (bf16_t)s[2 * u][e];
                    pf[4 + e] = (bf16_t)s[2 * u + 1][e];
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const lds_char* base = vfrag[dt] + 32 * 128 * u;
                    const s16x4 lo = lds_tr16(base);
                    const s16x4 hi = lds_tr16(base + 16 * 128);
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[dt], 0, 0, 0);
                }
            }
        }
#ifdef AFWD_STAMPS
        asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
#endif
        // ---- store: lane owns query qrow, d = 16dt + 4g + r
        if (qrow < L) {
            bf16_t* orow = (bf16_t*)p.out + (row0 + qrow) * p.ldo + (size_t)h * 64 + 4 * g;
            // K-blocked output (A operand of the output projection): column block 2h + (dt >> 1), 16·(dt & 1) + 4g inside it
            bf16_t* okb = (bf16_t*)p.out + ((size_t)(2 * h) * (size_t)p.out_kb_rows + row0 + qrow) * 32 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
                store4(p.out_kb_rows ? okb + (size_t)(dt >> 1) * (size_t)p.out_kb_rows * 32 + 16 * (dt & 1) : orow + 16 * dt, v);
            }
            if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * L + qrow] = (msc + log2f(sum)) * 0.69314718055994531f;
        }
#ifdef AFWD_STAMPS
        const unsigned long long st4 = __builtin_amdgcn_s_memtime();
        st_acc[0] += st1 - st0; st_acc[1] += st2 - st1; st_acc[2] += st3 - st2; st_acc[3] += st4 - st3;
#endif
    }
#ifdef AFWD_STAMPS
    if (blockIdx.x == 1500 && lane == 0 && wave < 8) {
        unsigned long long* o = uia_afwd_stamps + wave * 8;
        o[0] = st_issued - st_begin; o[1] = st_staged - st_issued; o[2] = st_acc[0]; o[3] = st_acc[1]; o[4] = st_acc[2]; o[5] = st_acc[3];
        o[6] = __builtin_amdgcn_s_memtime() - st_begin;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// fp32 parity path: one query per thread, K/V rows broadcast from LDS.
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const int L = p.cu_seqlens ? p.cu_seqlens[b + 1] - p.cu_seqlens[b] : p.L;       // packed sequences: see the bf16 kernel
    float* Ks = (float*)smem;               // [L][64]
    float* Vs = Ks + (size_t)L * 64;
    const int tid = threadIdx.x;
    const size_t row0 = p.cu_seqlens ? (size_t)p.cu_seqlens[b] : (size_t)b * L;
    const float* qb = (const float*)p.q + row0 * p.ld_qkv + (size_t)h * 64;
    const float* kb = (const float*)p.k + row0 * p.ld_qkv + (size_t)h * 64;
    const float* vb = (const float*)p.v + row0 * p.ld_qkv + (size_t)h * 64;
    for (int i = tid; i < L * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        *(f32x4*)(Ks + r * 64 + c) = *(const f32x4*)(kb + (size_t)r * p.ld_qkv + c);
        *(f32x4*)(Vs + r * 64 + c) = *(const f32x4*)(vb + (size_t)r * p.ld_qkv + c);
    }
    __syncthreads();
    int klen = L;
    if (p.mask_kind == UIA_MASK_KEYPAD && p.keylen) { klen = p.keylen[b]; klen = klen < 1 ? 1 : (klen > L ? L : klen); }
    for (int qi = tid; qi < L; qi += 256) {
        float q[64];
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            const f32x4 v = *(const f32x4*)(qb + (size_t)qi * p.ld_qkv + c);
            q[c] = v[0] * p.scale; q[c + 1] = v[1] * p.scale; q[c + 2] = v[2] * p.scale; q[c + 3] = v[3] * p.scale;
        }
        const int kend = p.mask_kind == UIA_MASK_CAUSAL ? (qi + 1 < klen ? qi + 1 : klen) : klen;
        float m = -INFINITY;
        for (int k = 0; k < kend; ++k) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) s = fmaf(q[c], Ks[k * 64 + c], s);
            m = fmaxf(m, s);
        }
        float o[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) o[c] = 0.f;
        float sum = 0.f;
        for (int k = 0; k < kend; ++k) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) s = fmaf(q[c], Ks[k * 64 + c], s);
            const float e = expf(s - m);
            sum += e;
#pragma unroll
            for (int c = 0; c < 64; ++c) o[c] = fmaf(e, Vs[k * 64 + c], o[c]);
        }
        const float inv = 1.0f / sum;
#ifdef AFWD_STAMPS
        asm volatile("" :: "v"(inv));
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
        float* orow = (float*)p.out + (row0 + qi) * p.ldo + (size_t)h * 64;
#pragma unroll
        for (int c = 0; c < 64; c += 4) *(f32x4*)(orow + c) = f32x4{o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
        if (p.lse) p.lse[((size_t)b * p.H + h) * L + qi] = m + logf(sum);
    }
}

template <int LT_MAX, int NW>
int launch_bf16(hipStream_t stream, const UiaAttnParams& p) {
    const int LT = (p.L + 15) / 16, NP = (LT + 1) / 2;
    const int lds = ((LT + 3) / 4) * 4 * 2048 + NP * 32 * 128;          // K in whole chunks of four tiles, V in pairs
    auto kern = attn_fwd_bf16_kernel<LT_MAX, NW>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, ((LT_MAX + 3) / 4) * 4 * 2048 + ((LT_MAX + 1) / 2) * 32 * 128);
    if (getenv("UIA_ATTN_FWD_OCC")) {           // diagnostic: workgroups of this launch a CU can hold at once
        int per_cu = -1;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, lds);
        fprintf(stderr, "attn_fwd<%d,%d>: %d threads, %d B of LDS -> %d workgroup(s) per CU (%s)\n", LT_MAX, NW, 64 * NW, lds, per_cu, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(64 * NW), lds, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int uia_attn_fwd_launch(hipStream_t stream, int dtype, const UiaAttnParams& p) {
    UIA_CHECK_ARG(dtype == UIA_BF16 || dtype == UIA_F32, "uia_attn_fwd: bad dtype %d", dtype);
    UIA_CHECK_ARG(p.B > 0 && p.H > 0 && p.L > 0, "uia_attn_fwd: empty problem");
    UIA_CHECK_ARG(p.scale > 0.f && p.scale < 3.0e38f, "uia_attn_fwd: scale must be positive and finite (the row max is taken on the raw scores), got %g", (double)p.scale);
    UIA_CHECK_ARG((p.out_kb_rows == 0 && p.dqkv_kb_rows == 0) || (p.dh == 64 && dtype == UIA_BF16 && p.dqkv_kb_rows == 0 && p.out_kb_rows >= (int64_t)p.B * p.L),
                  "uia_attn_fwd: a K-blocked output needs the bf16 head-dim-64 path and out_kb_rows >= B*L");
    if (p.dh != 64) return uia_attn_small_launch(stream, dtype, p, false);   // CLIPSeg decoder heads (d_h = 16)
    UIA_CHECK_ARG(p.L <= 272, "uia_attn_fwd: L=%d exceeds the single-pass limit 272", p.L);
    UIA_CHECK_ARG(p.q && p.k && p.v && p.out, "uia_attn_fwd: null tensor");
    const int esz = dtype == UIA_BF16 ? 2 : 4;
    UIA_CHECK_ARG((p.ld_qkv * esz) % 16 == 0 && (p.ldo * esz) % 8 == 0, "uia_attn_fwd: leading dimensions must keep 16-byte rows");
    UIA_CHECK_ARG(((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v) % 16 == 0 && (uintptr_t)p.out % 8 == 0, "uia_attn_fwd: alignment");
    UIA_CHECK_ARG(p.mask_kind >= UIA_MASK_NONE && p.mask_kind <= UIA_MASK_KEYPAD, "uia_attn_fwd: bad mask kind");
    UIA_CHECK_ARG(p.mask_kind != UIA_MASK_KEYPAD || p.keylen, "uia_attn_fwd: key-padding mask needs keylen");
    UIA_CHECK_ARG(!p.cu_seqlens || (p.mask_kind != UIA_MASK_KEYPAD && !p.lse),
                  "uia_attn_fwd: packed sequences (cu_seqlens) take no key-padding mask and write no lse (forward-only path)");
    if (dtype == UIA_F32) {
        const int lds = 2 * p.L * 64 * 4;
        static UiaDevOnce attr_once;
        UIA_ENSURE_LDS_ATTR(attr_once, attn_fwd_f32_kernel, 2 * 272 * 64 * 4);
        hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(p.B * p.H), dim3(256), lds, stream, p);
        UIA_CHECK_LAUNCH();
        return 0;
    }
    const int LT = (p.L + 15) / 16;
    if (LT <= 5) return launch_bf16<5, 4>(stream, p);
    if (LT <= 13) return launch_bf16<13, 7>(stream, p);
    if (LT <= 16) return launch_bf16<16, 8>(stream, p);      // BERT's 256 tokens: four chunks of four key tiles, no fifth (the 17-tile instantiation spills ten registers at 128)
    return launch_bf16<17, 8>(stream, p);
}
