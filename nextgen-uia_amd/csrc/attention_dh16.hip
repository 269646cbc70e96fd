// attention_dh16.hip — softmax(q kᵀ·scale) v and its backward for head dim 16 on the matrix cores (bf16, no mask, L <= 512).
//
// The CLIPSeg decoder's three post-LN layers (reduce_dim 64, four heads of 16, 22 x 22 + 1 = 485 tokens; reference
// src/third_party/openai_clip/clipseg_adapter.py:73-98 driving transformers' CLIPSegDecoder) ran on decoder.hip's scalar kernels — one query per
// thread, K / V broadcast from LDS, ~130 FMAs per (query, key) pair: 70 us forward and 206 us backward per layer at 512 heads, all of it VALU issue.
// Head dim 16 is exactly the K of v_mfma_f32_16x16x16_bf16: one MFMA per 16 x 16 tile of S, and one per tile for each product that consumes P or dS.
//
// Forward (four waves per head, K row-major and Vᵀ in LDS): a wave owns a 16-query tile, computes Sᵀ = K·Qᵀ for ALL key tiles into registers (lane: query
// li, keys 16t + 4g + r — the whole softmax row of a query sits in four lanes), reduces max and sum with two lane exchanges, and feeds the normalised P
// straight back as the A operand of O = P·V (the Sᵀ accumulator layout IS the A layout of the next product).
// Backward (eight waves per head; Q, K, V, dO row-major and Qᵀ, Kᵀ, dOᵀ transposed in LDS, lse and δ beside them): the units of the head-dim-64 kernel
// (attention_bwd.hip) in small — KEY(kt): dK, dV of one key tile over all query tiles from S and dP; QRY(qt): dQ of one query tile over all key tiles from
// Sᵀ and dPᵀ — pulled from an LDS counter, no barrier after the staging, every product's second operand read as one aligned 8-byte LDS word.
// P and dS are rounded to bf16 as MFMA operands (the scalar kernels kept them in fp32): same rounding sites as the head-dim-64 kernels.
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int DH = 16, LT_MAX = 32;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
    typedef bf16_t b4 __attribute__((ext_vector_type(4)));
    const b4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    return __builtin_bit_cast(s16x4, v);
}
__device__ __forceinline__ s16x4 lds8(const char* p) { return *(const s16x4*)p; }

// row-major image [LP][16] bf16 (32-byte rows) of columns h·16 .. of `src`; rows >= L are zero
__device__ __forceinline__ void stage_rows(char* dst, const bf16_t* src, long ld, int L, int LP, int tid, int nthr) {
    for (int c = tid; c < LP * 2; c += nthr) {
        const int r = c >> 1, half = c & 1;
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (r < L) v = *(const uint4*)(src + (size_t)r * ld + 8 * half);
        *(uint4*)(dst + r * 32 + half * 16) = v;
    }
}
// transposed image [16][LPS] bf16 from the row-major LDS image (after a barrier)
__device__ __forceinline__ void stage_transposed(char* dst, const char* rows, int LP, int LPS, int tid, int nthr) {
    for (int c = tid; c < LP * 4; c += nthr) {
        const int r = c >> 2, q = c & 3;
        const s16x4 v = lds8(rows + r * 32 + q * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) *(short*)(dst + ((4 * q + e) * LPS + r) * 2) = v[e];
    }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_dh16_fwd_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = p.L, LT = (L + 15) >> 4, LP = LT * 16, LPS = LP + 4;
    char* Kr = smem;                                   // [LP][16]
    char* Vr = Kr + LP * 32;                           // [LP][16] (staging only)
    char* Vt = Vr + LP * 32;                           // [16][LPS]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const size_t row0 = (size_t)b * L;
    const bf16_t* qb = (const bf16_t*)p.q + row0 * p.ld_qkv + (size_t)h * DH;
    const bf16_t* kb = (const bf16_t*)p.k + row0 * p.ld_qkv + (size_t)h * DH;
    const bf16_t* vb = (const bf16_t*)p.v + row0 * p.ld_qkv + (size_t)h * DH;
    stage_rows(Kr, kb, p.ld_qkv, L, LP, tid, 256);
    stage_rows(Vr, vb, p.ld_qkv, L, LP, tid, 256);
    __syncthreads();
    stage_transposed(Vt, Vr, LP, LPS, tid, 256);
    __syncthreads();
    const float c2 = p.scale * LOG2E;
    for (int qt = wave; qt < LT; qt += 4) {
        const int qrow = 16 * qt + li;
        s16x4 qf = s16x4{0, 0, 0, 0};
        if (qrow < L) qf = *(const s16x4*)(qb + (size_t)qrow * p.ld_qkv + 4 * g);
        f32x4 s[LT_MAX];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < LT_MAX; ++t) {
            if (t < LT) {
                s[t] = mfma16(lds8(Kr + (16 * t + li) * 32 + 8 * g), qf, f32x4{0.f, 0.f, 0.f, 0.f});      // lane: key 16t + 4g + r, query li
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][r] = 16 * t + 4 * g + r < L ? s[t][r] * c2 : -INFINITY;
                    m = fmaxf(m, s[t][r]);
                }
            }
        }
        m = rows_max(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < LT_MAX; ++t) {
            if (t < LT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][r] = exp2f(s[t][r] - m); sum += s[t][r]; }
            }
        }
        sum = rows_sum(sum);
        const float inv = 1.0f / sum;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < LT_MAX; ++t) {
            if (t < LT) o = mfma16(pack4(s[t][0] * inv, s[t][1] * inv, s[t][2] * inv, s[t][3] * inv), lds8(Vt + (li * LPS + 16 * t + 4 * g) * 2), o);   // lane: query 4g + r, dh li
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * qt + 4 * g + r;
            if (q < L) ((bf16_t*)p.out)[(row0 + q) * p.ldo + (size_t)h * DH + li] = (bf16_t)o[r];
        }
        if (p.lse && g == 0 && qrow < L) p.lse[((size_t)b * p.H + h) * L + qrow] = (m + log2f(sum)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(512) void attn_dh16_bwd_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = p.L, LT = (L + 15) >> 4, LP = LT * 16, LPS = LP + 4;
    char* Qr = smem;
    char* Kr = Qr + LP * 32;
    char* Vr = Kr + LP * 32;
    char* Gr = Vr + LP * 32;                           // dO
    char* Qt = Gr + LP * 32;                           // [16][LPS]
    char* Kt = Qt + 16 * LPS * 2;
    char* Gt = Kt + 16 * LPS * 2;
    float* lse2 = (float*)(Gt + 16 * LPS * 2);         // [LP]: lse·log2e, +inf on the padding rows
    float* dlt = lse2 + LP;                            // [LP]: δ = Σ dO·O
    int* queue = (int*)(dlt + LP);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const size_t row0 = (size_t)b * L;
    const bf16_t* qb = (const bf16_t*)p.q + row0 * p.ld_qkv + (size_t)h * DH;
    const bf16_t* kb = (const bf16_t*)p.k + row0 * p.ld_qkv + (size_t)h * DH;
    const bf16_t* vb = (const bf16_t*)p.v + row0 * p.ld_qkv + (size_t)h * DH;
    const bf16_t* gb = (const bf16_t*)p.dout + row0 * p.lddo + (size_t)h * DH;
    const bf16_t* ob = (const bf16_t*)p.out + row0 * p.ldo + (size_t)h * DH;
    stage_rows(Qr, qb, p.ld_qkv, L, LP, tid, 512);
    stage_rows(Kr, kb, p.ld_qkv, L, LP, tid, 512);
    stage_rows(Vr, vb, p.ld_qkv, L, LP, tid, 512);
    stage_rows(Gr, gb, p.lddo, L, LP, tid, 512);
    for (int r = tid; r < LP; r += 512) {
        float d = 0.f, l2 = INFINITY;
        if (r < L) {
            typedef bf16_t b8 __attribute__((ext_vector_type(8)));
            const b8 g0 = *(const b8*)(gb + (size_t)r * p.lddo), g1 = *(const b8*)(gb + (size_t)r * p.lddo + 8);
            const b8 o0 = *(const b8*)(ob + (size_t)r * p.ldo), o1 = *(const b8*)(ob + (size_t)r * p.ldo + 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf((float)g0[e], (float)o0[e], fmaf((float)g1[e], (float)o1[e], d));
            l2 = p.lse[((size_t)b * p.H + h) * L + r] * LOG2E;
        }
        dlt[r] = d;
        lse2[r] = l2;
    }
    if (tid == 0) *queue = 0;
    __syncthreads();
    stage_transposed(Qt, Qr, LP, LPS, tid, 512);
    stage_transposed(Kt, Kr, LP, LPS, tid, 512);
    stage_transposed(Gt, Gr, LP, LPS, tid, 512);
    __syncthreads();
    const float c2 = p.scale * LOG2E;
    for (;;) {
        int u = 0;
        if (lane == 0) u = atomicAdd(queue, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= 2 * LT) break;
        if (u < LT) {
            // ---- KEY unit: dK, dV of key tile kt.  S, dP: lane [query 4g + r][key li]
            const int kt = u, key = 16 * kt + li;
            const s16x4 kf = lds8(Kr + key * 32 + 8 * g), vf = lds8(Vr + key * 32 + 8 * g);
            const bool key_ok = key < L;
            f32x4 dkt = {0.f, 0.f, 0.f, 0.f}, dvt = {0.f, 0.f, 0.f, 0.f};                     // lane [dh 4g + r][key li]
            for (int qt = 0; qt < LT; ++qt) {
                const int qo = (16 * qt + li) * 32 + 8 * g;
                const f32x4 s = mfma16(lds8(Qr + qo), kf, f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 dp = mfma16(lds8(Gr + qo), vf, f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 l2 = *(const f32x4*)(lse2 + 16 * qt + 4 * g), dl = *(const f32x4*)(dlt + 16 * qt + 4 * g);
                float pv[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = key_ok ? exp2f(fmaf(s[r], c2, -l2[r])) : 0.f;                      // padding queries: lse2 = +inf -> 0
                    ds[r] = pv[r] * (dp[r] - dl[r]);
                }
                const int to = (li * LPS + 16 * qt + 4 * g) * 2;
                dvt = mfma16(lds8(Gt + to), pack4(pv[0], pv[1], pv[2], pv[3]), dvt);
                dkt = mfma16(lds8(Qt + to), pack4(ds[0], ds[1], ds[2], ds[3]), dkt);
            }
            if (key_ok) {
                const size_t o = (row0 + key) * p.ld_dqkv + (size_t)h * DH + 4 * g;
                *(s16x4*)((bf16_t*)p.dv + o) = pack4(dvt[0], dvt[1], dvt[2], dvt[3]);
                *(s16x4*)((bf16_t*)p.dk + o) = pack4(dkt[0] * p.scale, dkt[1] * p.scale, dkt[2] * p.scale, dkt[3] * p.scale);
            }
        } else {
            // ---- QRY unit: dQ of query tile qt.  Sᵀ, dPᵀ: lane [key 4g + r][query li]
            const int qt = u - LT, qrow = 16 * qt + li;
            const s16x4 qf = lds8(Qr + qrow * 32 + 8 * g), gf = lds8(Gr + qrow * 32 + 8 * g);
            const float l2 = lse2[qrow], dl = dlt[qrow];
            f32x4 dqt = {0.f, 0.f, 0.f, 0.f};                                                  // lane [dh 4g + r][query li]
            for (int kt = 0; kt < LT; ++kt) {
                const int ko = (16 * kt + li) * 32 + 8 * g;
                const f32x4 s = mfma16(lds8(Kr + ko), qf, f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 dp = mfma16(lds8(Vr + ko), gf, f32x4{0.f, 0.f, 0.f, 0.f});
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = 16 * kt + 4 * g + r < L ? exp2f(fmaf(s[r], c2, -l2)) : 0.f;
                    ds[r] = pv * (dp[r] - dl);
                }
                dqt = mfma16(lds8(Kt + (li * LPS + 16 * kt + 4 * g) * 2), pack4(ds[0], ds[1], ds[2], ds[3]), dqt);
            }
            if (qrow < L)
                *(s16x4*)((bf16_t*)p.dq + (row0 + qrow) * p.ld_dqkv + (size_t)h * DH + 4 * g) = pack4(dqt[0] * p.scale, dqt[1] * p.scale, dqt[2] * p.scale, dqt[3] * p.scale);
        }
    }
}

}  // namespace

bool uia_attn_dh16_ok(int dtype, const UiaAttnParams& p) {
    return dtype == UIA_BF16 && p.dh == 16 && p.L <= 16 * LT_MAX && p.mask_kind == UIA_MASK_NONE && !p.cu_seqlens && p.out_kb_rows == 0 && p.dqkv_kb_rows == 0 &&
           p.ld_qkv % 8 == 0 && p.ldo % 8 == 0 && ((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.out) % 16 == 0;
}

int uia_attn_dh16_launch(hipStream_t stream, const UiaAttnParams& p, bool bwd) {
    const int LT = (p.L + 15) / 16, LP = LT * 16, LPS = LP + 4;
    if (!bwd) {
        const int lds = 2 * LP * 32 + 16 * LPS * 2;
        hipLaunchKernelGGL(attn_dh16_fwd_kernel, dim3(p.B * p.H), dim3(256), lds, stream, p);
    } else {
        UIA_CHECK_ARG(p.dout && p.lse && p.dq && p.dk && p.dv && p.lddo % 8 == 0 && p.ld_dqkv % 4 == 0 && (uintptr_t)p.dout % 16 == 0 &&
                          ((uintptr_t)p.dq | (uintptr_t)p.dk | (uintptr_t)p.dv) % 8 == 0,
                      "uia_attn_bwd (head dim 16): dout, lse, dq, dk, dv with 16-byte aligned dout rows and 8-byte aligned gradient rows");
        const int lds = 4 * LP * 32 + 3 * 16 * LPS * 2 + 2 * LP * 4 + 16;
        static UiaDevOnce once;
        UIA_ENSURE_LDS_ATTR(once, attn_dh16_bwd_kernel, 4 * 512 * 32 + 3 * 16 * 516 * 2 + 2 * 512 * 4 + 16);
        hipLaunchKernelGGL(attn_dh16_bwd_kernel, dim3(p.B * p.H), dim3(512), lds, stream, p);
    }
    UIA_CHECK_LAUNCH();
    return 0;
}
