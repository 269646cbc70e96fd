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

}  // namespace

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
