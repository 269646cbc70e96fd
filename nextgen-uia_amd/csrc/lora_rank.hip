// lora_rank.hip — uia_lora_rank_update: out += Σ_i drop_i( alpha · Q_i · W_iᵀ ), up to three rank-64 sources in ONE read-modify-write pass.
//
// The data gradient of a fused q | k | v projection with LinearLoRA on each part (reference src/adapters/lora.py:78-90 inside an OpenAI-CLIP residual
// block): dh = dqkv·[Wq; Wk; Wv] + Σ_i mask_i ⊙ (s·(dy_i·B_i)·A_i) / (1 - p).  Every LinearLoRA owns its nn.Dropout, so the three masks differ and the
// rank terms cannot ride in the frozen GEMM's K loop as they do in the forward (the mask multiplies the rank term only).  Round 3 ran one K = 64 stream
// launch (tile cfg 23) per source: three read-modify-write passes over dh (ViT-L/14, batch 128: 3 x 134 MB, 3 x 39 us per layer).  This kernel makes it
// one pass: 2·M·N·64·n FLOP against 4·M·N bytes — still a stream over the result.
//
// Layout: a workgroup owns a QUARTER of the columns (N/4, a multiple of 64) and keeps the n sources' [N/4 x 64] weight rows in LDS (3 x 256 x 144 B =
// 108 KB at N = 1024); a wave's unit is 16 rows x 64 columns: its n x 2 A fragments come straight from global memory (the Q operand is n·M·128 bytes:
// L2), the weight rows are read from LDS in the order that leaves a lane 16 consecutive columns of one row (as tile cfg 23 does), the masks are drawn
// per eight columns from (seed_i, element index / 8) exactly as uia_dropout / drop_where = 2 draw them, and the lane adds the sum to its 32 bytes of the
// result.  The next unit's operands are requested before the current one is multiplied.
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

constexpr int LDWB = 128 + 16;                             // row stride of a weight image: +16 B staggers the rows over the banks

template <int NSRC>
__global__ __launch_bounds__(512) void lora_rank_update_kernel(const uia_lora_rank_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = p.N >> 2;                               // columns of this workgroup's quarter
    const int qc = blockIdx.x & 3, n_base = qc * nq;
    const int nch = nq >> 6;                               // 64-column chunks per quarter
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const char* wsrc = (const char*)p.W[s] + ((size_t)n_base * p.ldw) * 2;
        for (int c = tid; c < nq * 8; c += 512) {
            const int r = c >> 3, cc = c & 7;
            *(uint4*)(smem + (s * nq + r) * LDWB + cc * 16) = *(const uint4*)(wsrc + ((size_t)r * p.ldw) * 2 + cc * 16);
        }
    }
    __syncthreads();
    const int ntiles = (p.M + 15) >> 4;
    const int nunits = ntiles * nch, ustep = (gridDim.x >> 2) * 8;
    const bool drop = p.drop_p > 0.f;
    const uint32_t drop_th = dropout_thresh16(p.drop_p);
    const float drop_inv = 1.0f / (1.0f - p.drop_p);
    bf16_t* out = (bf16_t*)p.out;
    // weight fragment of (source s, chunk c, tile nt, k-step ks): row 64c + 16(li>>2) + 4nt + (li&3), bytes 64ks + 16g
    const char* wfrag = smem + (16 * (li >> 2) + (li & 3)) * LDWB + g * 16;

    struct Unit { uint4 a[NSRC][2]; uint4 r0, r1; };
    auto request = [&](int u, Unit& q) {
        const int rt = u / nch, c = u - rt * nch;
        const int m = 16 * rt + li, mc = m < p.M ? m : p.M - 1;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const char* arow = (const char*)p.Q + ((size_t)s * p.q_stride + (size_t)mc * p.ldq) * 2 + g * 16;
            q.a[s][0] = *(const uint4*)arow;
            q.a[s][1] = *(const uint4*)(arow + 64);
        }
        const bf16_t* rr = out + (size_t)mc * p.ldo + n_base + 64 * c + 16 * g;
        q.r0 = *(const uint4*)rr;
        q.r1 = *(const uint4*)(rr + 8);
    };
    auto process = [&](int u, const Unit& q) {
        const int rt = u / nch, c = u - rt * nch;
        const int m = 16 * rt + li;
        const int n = n_base + 64 * c + 16 * g;
        float v[16];
        {
            const bf16x8 x0 = __builtin_bit_cast(bf16x8, q.r0), x1 = __builtin_bit_cast(bf16x8, q.r1);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = (float)x0[e]; v[8 + e] = (float)x1[e]; }
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const char* wc = wfrag + (size_t)(s * nq + 64 * c) * LDWB;
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const uint4 w0 = *(const uint4*)(wc + (4 * nt) * LDWB), w1 = *(const uint4*)(wc + (4 * nt) * LDWB + 64);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, q.a[s][0]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, q.a[s][1]), acc[nt], 0, 0, 0);
            }
            uint32_t k0 = 0xFFu, k1 = 0xFFu;
            float sc = p.alpha;
            if (drop) {
                const uint32_t grp = (uint32_t)(((size_t)(m < p.M ? m : 0) * (size_t)p.N + (size_t)n) >> 3);
                k0 = dropout_keep8(p.seed[s], grp, drop_th);
                k1 = dropout_keep8(p.seed[s], grp + 1, drop_th);
                sc = p.alpha * drop_inv;                   // (alpha·x)·inv and alpha·inv·x differ in the last bit only; the sum below is rounded to bf16 once
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 4 * nt + r;
                    const bool keep = ((e < 8 ? k0 >> e : k1 >> (e - 8)) & 1u) != 0;
                    v[e] += keep ? acc[nt][r] * sc : 0.f;
                }
        }
        if (m >= p.M) return;
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o0[e] = (bf16_t)v[e]; o1[e] = (bf16_t)v[8 + e]; }
        bf16_t* o = out + (size_t)m * p.ldo + n;
        *(bf16x8*)o = o0;
        *(bf16x8*)(o + 8) = o1;
    };
    // two units requested ahead of the one being multiplied (three register sets, the loop unrolled by three)
    int u = (blockIdx.x >> 2) * 8 + wave;
    Unit q0, q1, q2;
    if (u < nunits) request(u, q0);
    if (u + ustep < nunits) request(u + ustep, q1);
    while (u < nunits) {
        if (u + 2 * ustep < nunits) request(u + 2 * ustep, q2);
        process(u, q0);
        u += ustep;
        if (u >= nunits) break;
        if (u + 2 * ustep < nunits) request(u + 2 * ustep, q0);
        process(u, q1);
        u += ustep;
        if (u >= nunits) break;
        if (u + 2 * ustep < nunits) request(u + 2 * ustep, q1);
        process(u, q2);
        u += ustep;
    }
}

template <int NSRC>
int launch_rank(hipStream_t stream, const uia_lora_rank_desc& p) {
    const int lds = NSRC * (p.N / 4) * LDWB;
    static UiaDevOnce once;
    UIA_ENSURE_LDS_ATTR(once, lora_rank_update_kernel<NSRC>, 160 * 1024);
    const int ncu = uia_num_cus();
    const long units = (long)((p.M + 15) / 16) * (p.N / 256);          // per column quarter
    long per_q = (units + 7) / 8;
    if (per_q > ncu / 4) per_q = ncu / 4;
    if (per_q < 1) per_q = 1;
    hipLaunchKernelGGL(lora_rank_update_kernel<NSRC>, dim3((unsigned)(4 * per_q)), dim3(512), lds, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// uia_ln_lora_down: h = LayerNorm(x) AND t_s = drop_s(h)·A_sᵀ for up to three LinearLoRA wrappers that share the input (q, k, v of a block) in ONE launch.
// Round 3 ran the LayerNorm kernel (x in, h out) and then one N = 64 stream launch per wrapper, each reading h back (67 MB at ViT-L/14, 128 pairs) to
// produce 16 useful columns: 30 + 3 x 28 us per block.  Here a block walks 16-row tiles: the four waves normalise four rows each (same association
// as ln_fwd_kernel: h is bit-identical) and leave them in global memory and in a bf16 LDS tile; then wave w multiplies its quarter of K (D/4 columns)
// of the tile with the three wrappers' rank rows — their fragments stay in registers for the whole launch — applying wrapper s's dropout mask to the
// tile fragment on the way (a 16-byte fragment is exactly one draw of dropout_keep8); the four K-partials meet in LDS.  Rank <= 16 (one MFMA row tile);
// columns 16..63 of t are written as zeros (the K-extension operand of the frozen GEMM is 64 wide).
#ifndef LNLD_MIN_WAVES
#define LNLD_MIN_WAVES 1
#endif
template <int NVF, int NSRC>                                   // D = 256·NVF
__global__ __launch_bounds__(256, LNLD_MIN_WAVES) void ln_lora_down_kernel(const uia_ln_lora_desc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 256 * NVF, ROWB = 2 * D + 16, KSW = D / 128;                     // k steps (32 columns) per wave
    f32x4* red = (f32x4*)(smem + 16 * ROWB);                                           // [4 waves][NSRC][64 lanes]
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 af[NSRC][KSW];                                       // A_s rank row li, columns 32(w·KSW + ks) + 8g .. +7
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) af[s][ks] = *(const uint4*)((const bf16_t*)p.A[s] + (size_t)li * p.lda + 32 * (wave * KSW + ks) + 8 * g);
    const bool drop = p.drop_p > 0.f;
    const uint32_t drop_th = dropout_thresh16(p.drop_p);
    const float drop_inv = 1.0f / (1.0f - p.drop_p);
    bf16_t* h = (bf16_t*)p.h;
    const int ntiles = (p.M + 15) >> 4;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = 16 * tile;
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int rl = 4 * wave + i, row = r0 + rl;
            char* hrow = smem + rl * ROWB;
            if (row >= p.M) {
#pragma unroll
                for (int k = 0; k < NVF; ++k) *(uint2*)(hrow + 8 * (lane + 64 * k)) = uint2{0u, 0u};
                continue;
            }
            const float* xr = p.x + (size_t)row * p.ldx;
            f32x4 v[NVF];
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NVF; ++k) { v[k] = load4(xr + 4 * (lane + 64 * k)); sm += v[k][0] + v[k][1] + v[k][2] + v[k][3]; }
            const float mean = wave_sum(sm) / D;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NVF; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mean; q = fmaf(d, d, q); }
            const float rstd = rsqrtf(wave_sum(q) / D + p.eps);
#pragma unroll
            for (int k = 0; k < NVF; ++k) {
                const int c = lane + 64 * k;
                const f32x4 gm = load4(p.gamma + 4 * c), b = load4(p.beta + 4 * c);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaf((v[k][e] - mean) * rstd, gm[e], b[e]);
                store4(h + (size_t)row * D + 4 * c, y);
                store4((bf16_t*)(hrow + 8 * c), y);
            }
        }
        __syncthreads();
        f32x4 acc[NSRC];
#pragma unroll
        for (int s = 0; s < NSRC; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mrow = r0 + li < p.M ? r0 + li : p.M - 1;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            const int col = 32 * (wave * KSW + ks) + 8 * g;
            const uint4 hf = *(const uint4*)(smem + li * ROWB + 2 * col);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                uint4 b = hf;
                if (drop) {
                    const uint32_t keep = dropout_keep8(p.seed[s], (uint32_t)(((size_t)mrow * (size_t)D + (size_t)col) >> 3), drop_th);
                    bf16x8 v = __builtin_bit_cast(bf16x8, hf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((keep >> e) & 1u ? (float)v[e] * drop_inv : 0.f);     // as the N = 64 stream kernel applies it
                    b = __builtin_bit_cast(uint4, v);
                }
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[s][ks]), __builtin_bit_cast(bf16x8, b), acc[s], 0, 0, 0);   // D[rank 4g + r][row li]
            }
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) red[(wave * NSRC + s) * 64 + lane] = acc[s];
        __syncthreads();
        if (wave < NSRC && r0 + li < p.M) {
            const f32x4 a0 = red[(0 * NSRC + wave) * 64 + lane], a1 = red[(1 * NSRC + wave) * 64 + lane], a2 = red[(2 * NSRC + wave) * 64 + lane],
                        a3 = red[(3 * NSRC + wave) * 64 + lane];
            const f32x4 sum = (a0 + a1) + (a2 + a3);
            bf16_t* trow = (bf16_t*)p.T + (size_t)wave * p.t_stride + (size_t)(r0 + li) * 64;
            store4(trow + 4 * g, sum);
            const uint2 z = uint2{0u, 0u};
            *(uint2*)(trow + 16 + 12 * g) = z;
            *(uint2*)(trow + 20 + 12 * g) = z;
            *(uint2*)(trow + 24 + 12 * g) = z;
        }
        __syncthreads();
    }
}

template <int NVF, int NSRC>
int launch_ln_lora(hipStream_t stream, const uia_ln_lora_desc& p) {
    constexpr int D = 256 * NVF;
    const int lds = 16 * (2 * D + 16) + 4 * NSRC * 64 * 16;
    auto kern = ln_lora_down_kernel<NVF, NSRC>;
    static UiaDevOnce once;
    UIA_ENSURE_LDS_ATTR(once, kern, 64 * 1024);
    const int ntiles = (p.M + 15) / 16;
    int blocks = ntiles, per_cu = 0;
    const int ncu = uia_num_cus();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) == hipSuccess && per_cu > 0 && per_cu * ncu < blocks) blocks = per_cu * ncu;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int uia_ln_lora_down_launch(hipStream_t stream, int dtype, const uia_ln_lora_desc& p) {
    UIA_CHECK_ARG(dtype == UIA_BF16, "uia_ln_lora_down: bf16 only (dtype %d)", dtype);
    UIA_CHECK_ARG(p.M > 0 && (p.D == 768 || p.D == 1024) && p.nsrc >= 1 && p.nsrc <= 3, "uia_ln_lora_down: M=%d, D=%d (768 or 1024), nsrc=%d (1..3)", p.M, p.D, p.nsrc);
    UIA_CHECK_ARG(p.x && p.gamma && p.beta && p.h && p.T && p.ldx >= p.D && p.ldx % 4 == 0 && p.lda >= p.D && p.lda % 8 == 0 && (p.nsrc == 1 || p.t_stride >= (int64_t)p.M * 64) &&
                      ((uintptr_t)p.x | (uintptr_t)p.gamma | (uintptr_t)p.beta | (uintptr_t)p.h | (uintptr_t)p.T) % 16 == 0 && p.t_stride % 4 == 0,
                  "uia_ln_lora_down: null tensor, or x / h / T not 16-byte aligned with row strides ldx >= D, lda >= D (multiple of 8), t_stride >= 64·M");
    for (int s = 0; s < p.nsrc; ++s) UIA_CHECK_ARG(p.A[s] && (uintptr_t)p.A[s] % 16 == 0, "uia_ln_lora_down: A[%d] must be a 16-byte aligned bf16 [>= 16, D] matrix", s);
    UIA_CHECK_ARG(p.drop_p >= 0.f && p.drop_p < 1.f && (p.drop_p == 0.f || (size_t)p.M * (size_t)p.D / 8 <= 0xFFFFFFFFull), "uia_ln_lora_down: drop_p=%f", (double)p.drop_p);
    if (p.D == 768) {
        switch (p.nsrc) { case 1: return launch_ln_lora<3, 1>(stream, p); case 2: return launch_ln_lora<3, 2>(stream, p); default: return launch_ln_lora<3, 3>(stream, p); }
    }
    switch (p.nsrc) { case 1: return launch_ln_lora<4, 1>(stream, p); case 2: return launch_ln_lora<4, 2>(stream, p); default: return launch_ln_lora<4, 3>(stream, p); }
}

int uia_lora_rank_update_launch(hipStream_t stream, int dtype, const uia_lora_rank_desc& p) {
    UIA_CHECK_ARG(dtype == UIA_BF16, "uia_lora_rank_update: bf16 only (dtype %d)", dtype);
    UIA_CHECK_ARG(p.M > 0 && p.N > 0 && p.N % 256 == 0 && p.nsrc >= 1 && p.nsrc <= 3, "uia_lora_rank_update: M=%d, N=%d (a multiple of 256), nsrc=%d (1..3)", p.M, p.N, p.nsrc);
    UIA_CHECK_ARG(p.nsrc * (p.N / 4) * LDWB <= 160 * 1024, "uia_lora_rank_update: %d sources of N/4 = %d weight rows do not fit the LDS", p.nsrc, p.N / 4);
    UIA_CHECK_ARG(p.Q && p.out && (uintptr_t)p.Q % 16 == 0 && (uintptr_t)p.out % 16 == 0 && p.ldq >= 64 && p.ldq % 8 == 0 && p.ldo >= p.N && p.ldo % 8 == 0 &&
                      (p.nsrc == 1 || (p.q_stride >= (int64_t)p.M * p.ldq && p.q_stride % 8 == 0)),
                  "uia_lora_rank_update: Q is bf16 [nsrc][M][64] (ldq >= 64, q_stride >= M·ldq), out bf16 [M][N]; both 16-byte aligned with leading dimensions in multiples of 8");
    for (int s = 0; s < p.nsrc; ++s)
        UIA_CHECK_ARG(p.W[s] && (uintptr_t)p.W[s] % 16 == 0, "uia_lora_rank_update: W[%d] must be a 16-byte aligned bf16 [N][64] matrix", s);
    UIA_CHECK_ARG(p.ldw >= 64 && p.ldw % 8 == 0, "uia_lora_rank_update: ldw=%lld", (long long)p.ldw);
    UIA_CHECK_ARG(p.drop_p >= 0.f && p.drop_p < 1.f, "uia_lora_rank_update: drop_p=%f outside [0, 1)", (double)p.drop_p);
    UIA_CHECK_ARG(p.drop_p == 0.f || (size_t)p.M * (size_t)p.N / 8 <= 0xFFFFFFFFull, "uia_lora_rank_update: dropout needs M*N <= 2^35");
    switch (p.nsrc) {
        case 1: return launch_rank<1>(stream, p);
        case 2: return launch_rank<2>(stream, p);
        default: return launch_rank<3>(stream, p);
    }
}
