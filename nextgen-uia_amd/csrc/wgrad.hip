// wgrad.hip — parameter gradient of a Linear:  dW[I,J] += alpha · Σ_m A[m,I]ᵀ · B[m,J]
//            (+ optional dbias[I] += Σ_m A[m,I]).   I and J multiples of 64.
//
// The reference gets these from autograd; only ADAPTER weights are trainable on the hot path, so
// one of I, J is always the bottleneck / rank (64):
//     Mona  dW2 = dYᵀ·d, db2 = Σ dY     (project2, /root/reference/src/adapters/mona.py:148,358)
//     Mona  dW1 = dtᵀ·u, db1 = Σ dt     (project1, mona.py:127,331)
//     LoRA  dB = s·dyᵀ·t, dA = s·qᵀ·x̃, db = Σ dy    (src/adapters/lora.py:87)
//
// The contraction runs over the ROW index of both row-major operands, so both MFMA fragments
// need "k = rows": they are fetched from 128-row LDS slabs with ds_read_b64_tr_b16 (bf16) or with
// plain 4-byte reads for the 16x16x4 fp32 MFMA.  A workgroup owns one 64×64 tile of dW and one
// chunk of rows; its four waves split each slab by rows, are summed through LDS, and the tile is
// accumulated into the (caller-zeroed) fp32 gradient with one atomic add per element per chunk.
// HBM-bound (reads A once per J-tile and B once per I-tile).
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

constexpr int SLAB = 128;            // rows per LDS slab (32 per wave)
#ifndef WG_CHUNK_SLABS
#define WG_CHUNK_SLABS 4
#endif
constexpr int CHUNK_SLABS = WG_CHUNK_SLABS;       // slabs per workgroup → 512 rows per chunk

// ---- bf16: slabs are [128][128 B] images, 32-B column groups swizzled by (row>>1)&3
// DROPB (LoRA's dA = s·qᵀ·drop(x), lora.py:82-87): B holds the UN-dropped rows; the mask uia_dropout / uia_gemm's drop_where = 1 draw for the [M, drop_ld]
// tensor B is a column window of is regenerated from (seed, element index / 8) while the slab is staged — a lane's 16-byte piece is exactly one draw —
// so the forward does not have to write the dropped rows out for this launch (67 MB per LinearLoRA at ViT-L/14, 128 pairs).
#ifndef WG_MIN_WAVES
#define WG_MIN_WAVES 1
#endif
// Up to four problems of one shape per launch (uia_wgrad_group: the q | k | v factors of a LoRA block — three launches of ~1200 short workgroups each were three
// latencies; blockIdx.y picks the problem).
struct WgradGroups {
    const bf16_t* A[4];
    const bf16_t* B[4];
    float* dW[4];
    float* dbias[4];
    unsigned long long seed[4];
};

template <bool DROPB>
__global__ __launch_bounds__(256, DROPB ? 3 : WG_MIN_WAVES) void wgrad_bf16_kernel(int M, int I, int J, const WgradGroups gp, long lda, long ldb, float alpha,
                                                          long ldw, int i_valid, int j_valid, float drop_p, long drop_ld, int drop_col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16_t* __restrict__ A = gp.A[blockIdx.y];
    const bf16_t* __restrict__ B = gp.B[blockIdx.y];
    float* __restrict__ dW = gp.dW[blockIdx.y];
    float* __restrict__ dbias = gp.dbias[blockIdx.y];
    const unsigned long long drop_seed = gp.seed[blockIdx.y];
    char* As = smem;                  // 16 KiB
    char* Bs = smem + SLAB * 128;     // 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (tile, row chunk), XCD-aware: blocks b and b + 8 share an XCD (round-robin dealing), and all tiles of ONE row chunk
    // go to one XCD, so the narrow operand's slab of that chunk (re-read by every tile of the wide operand: 12 of them for a 768 x 64
    // gradient) is fetched into one L2 instead of eight (fabric reads 130 -> ~85 MB per launch at M = 50 432; the L2 hit rate was 0.15).
    const int tiles = (I / 64) * (J / 64);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = xcd + 8 * (slot / tiles), tile = slot % tiles;
    const int tj = tile % (J / 64), ti = tile / (J / 64);
    const int m_begin = chunk * (SLAB * CHUNK_SLABS);
    if (m_begin >= M) return;                          // the grid is rounded up to whole groups of eight chunks
    const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool want_bias = dbias != nullptr && tj == 0;
    f32x4 bacc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};

    const uint32_t drop_th = dropout_thresh16(drop_p);
    const float drop_inv = 1.0f / (1.0f - drop_p);
    const int trow = 32 * wave + 4 * g + qq;          // lo rows; hi rows = +16
    const int tsw = (trow >> 1) & 3;                  // (+16 keeps (row>>1)&3)
    for (int sl = 0; sl < CHUNK_SLABS; ++sl) {
        const int m0 = m_begin + sl * SLAB;
        if (m0 >= M) break;
        __syncthreads();                               // previous slab fully consumed
        if constexpr (DROPB) {
            // all four pieces of the thread requested first, the four draws computed while they fly, then mask + ds_write (a plain load per piece in
            // line with its hash exposed one round trip per piece: 31 us per launch against 21 for the LDS-DMA form)
            uint4 raw[SLAB / 32];
            uint32_t keep[SLAB / 32];
#pragma unroll
            for (int i = 0; i < SLAB / 32; ++i) {
                const int q = wave + 4 * i, r = 8 * q + (lane >> 3);
                int gm = m0 + r;
                gm = gm < M ? gm : M - 1;
                const int c = ((lane & 7) ^ (((r >> 1) & 3) << 1)) * 16;
                glds16((const char*)(A + (size_t)gm * lda + (size_t)ti * 64) + c, As + q * 1024);
                raw[i] = *(const uint4*)((const char*)(B + (size_t)gm * ldb + (size_t)tj * 64) + c);
            }
#pragma unroll
            for (int i = 0; i < SLAB / 32; ++i) {
                const int q = wave + 4 * i, r = 8 * q + (lane >> 3);
                int gm = m0 + r;
                gm = gm < M ? gm : M - 1;
                const int c = ((lane & 7) ^ (((r >> 1) & 3) << 1)) * 16;
                keep[i] = dropout_keep8(drop_seed, (uint32_t)(((size_t)gm * (size_t)drop_ld + (size_t)(drop_col0 + tj * 64 + (c >> 1))) >> 3), drop_th);
            }
#pragma unroll
            for (int i = 0; i < SLAB / 32; ++i) {
                const int q = wave + 4 * i;
                bf16x8 v = __builtin_bit_cast(bf16x8, raw[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((keep[i] >> e) & 1u ? (float)v[e] * drop_inv : 0.f);     // what the N = 64 stream kernel writes, bit for bit
                *(bf16x8*)(Bs + q * 1024 + lane * 16) = v;
            }
        } else {
        for (int q = wave; q < SLAB / 8; q += 4) {
            const int r = 8 * q + (lane >> 3);
            int gm = m0 + r;
            gm = gm < M ? gm : M - 1;
            const int c = ((lane & 7) ^ (((r >> 1) & 3) << 1)) * 16;
            glds16((const char*)(A + (size_t)gm * lda + (size_t)ti * 64) + c, As + q * 1024);
            glds16((const char*)(B + (size_t)gm * ldb + (size_t)tj * 64) + c, Bs + q * 1024);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // this wave's 32 rows: k-slot (g,e) ↔ row 32w + 16(e>>2) + 4g + (e&3)
        const bool tail = m0 + SLAB > M;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = trow * 128 + ((t ^ tsw) << 5) + 8 * pp;
            s16x4 alo = lds_tr16(As + off), ahi = lds_tr16(As + off + 16 * 128);
            const s16x4 blo = lds_tr16(Bs + off), bhi = lds_tr16(Bs + off + 16 * 128);
            if (tail) {   // rows past M were clamped duplicates: zero their A contribution
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (m0 + 32 * wave + 4 * g + e >= M) alo[e] = 0;
                    if (m0 + 32 * wave + 16 + 4 * g + e >= M) ahi[e] = 0;
                }
            }
            const s16x8 av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            const s16x8 bv = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            af[t] = __builtin_bit_cast(bf16x8, av);
            bfr[t] = __builtin_bit_cast(bf16x8, bv);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        if (want_bias) {          // column sums of A = Aᵀ·1 on the matrix cores: four more MFMAs per slab (the first version read the slab
                                  // back with 32 two-byte LDS loads per thread: 17 of the 51 us at I = 768, tools/wgrad_variants.sh)
#pragma unroll
            for (int a = 0; a < 4; ++a) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], ones, bacc[a], 0, 0, 0);
        }
    }
    // ---- reduce the four waves through LDS in two rounds (waves 0,1 store; waves 2,3 add in place), then one atomic per element.
    //      The reduction buffer overlays the (dead) slabs and is no larger than they are: 32 KiB per workgroup, five workgroups per CU
    //      (a 64 KiB buffer for all four waves at once capped the kernel at two per CU: 48 us for 84 MB, latency-bound).
    __syncthreads();
    float* red = (float*)smem;                         // [2][64*64] fp32 = 32 KiB
    float* mine = red + (wave & 1) * 4096;
    if (wave < 2) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)            // D[row = i = 16a + 4g + r][col = j = 16b + li]
                    mine[(16 * a + 4 * g + r) * 64 + 16 * b + li] = acc[a][b][r];
    }
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine[(16 * a + 4 * g + r) * 64 + 16 * b + li] += acc[a][b][r];
    }
    __syncthreads();
#ifndef WG_NO_ATOMIC
    for (int e = tid; e < 4096; e += 256) {
        const float v = red[e] + red[4096 + e];
        const int i = ti * 64 + (e >> 6), j = tj * 64 + (e & 63);      // rows / columns past the valid extent are the caller's zero padding (LoRA rank)
        if (i < i_valid && j < j_valid) atomicAdd(dW + (size_t)i * ldw + j, v * alpha);
    }
#endif
    if (want_bias) {              // every column of bacc[a] holds Σ_m A[m][16a + 4g + r]: column 0 of each wave → LDS → one atomic per element
        __syncthreads();
        if (li == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 64 + 16 * a + 4 * g + r] = bacc[a][r];
        }
        __syncthreads();
        if (tid < 64 && ti * 64 + tid < i_valid) atomicAdd(dbias + ti * 64 + tid, (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]));
    }
}

// ---- fp32: 16x16x4 MFMA straight from global (parity path; not tuned)
__global__ __launch_bounds__(256) void wgrad_f32_kernel(int M, int I, int J, const float* __restrict__ A, long lda,
                                                         const float* __restrict__ B, long ldb, float alpha,
                                                         float* __restrict__ dW, float* __restrict__ dbias, long ldw, int i_valid, int j_valid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tj = blockIdx.x % (J / 64), ti = blockIdx.x / (J / 64);
    const int m_begin = blockIdx.y * (SLAB * CHUNK_SLABS);
    const int m_end = min(M, m_begin + SLAB * CHUNK_SLABS);
    const int li = lane & 15, g = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int m = m_begin + 4 * wave; m < m_end; m += 16) {
        const int row = m + g;
        const bool ok = row < m_end;
        const int rr = ok ? row : m_end - 1;
        float av[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            av[t] = ok ? A[(size_t)rr * lda + ti * 64 + 16 * t + li] : 0.f;
            bv[t] = B[(size_t)rr * ldb + tj * 64 + 16 * t + li];
            bsum += av[t];    // lane (li,g) accumulates columns 16t+li over its rows: summed per column below
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        (void)bsum;
    }
    __syncthreads();
    float* red = (float*)smem;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * 4096 + (16 * a + 4 * g + r) * 64 + 16 * b + li] = acc[a][b][r];
    __syncthreads();
    for (int e = tid; e < 4096; e += 256) {
        const float v = red[e] + red[4096 + e] + red[8192 + e] + red[12288 + e];
        const int i = ti * 64 + (e >> 6), j = tj * 64 + (e & 63);
        if (i < i_valid && j < j_valid) atomicAdd(dW + (size_t)i * ldw + j, v * alpha);
    }
    if (dbias && tj == 0) {   // separate simple pass: thread = column, quarter of the chunk's rows
        const int col = tid & 63, part = tid >> 6;
        float s = 0.f;
        for (int m = m_begin + part; m < m_end; m += 4) s += A[(size_t)m * lda + ti * 64 + col];
        __syncthreads();
        red[tid] = s;
        __syncthreads();
        if (tid < 64 && ti * 64 + tid < i_valid) atomicAdd(dbias + ti * 64 + tid, red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid]);
    }
}

}  // namespace

int uia_wgrad_group_launch(hipStream_t stream, int dtype, int n, int M, int I, int J, const void* const* A, long lda, const void* const* B, long ldb, float alpha,
                            float* const* dW, float* const* dbias, long ldw, int i_valid, int j_valid, float drop_p, const uint64_t* drop_seed, long drop_ld, int drop_col0) {
    if (ldw == 0) { ldw = J; i_valid = I; j_valid = J; }              // the plain form: a dense [I, J] gradient
    UIA_CHECK_ARG(dtype == UIA_BF16 || dtype == UIA_F32, "uia_wgrad: bad dtype %d", dtype);
    UIA_CHECK_ARG(n >= 1 && n <= 4 && (n == 1 || dtype == UIA_BF16), "uia_wgrad_group: 1..4 problems per launch (bf16), got %d", n);
    UIA_CHECK_ARG(M > 0 && I > 0 && J > 0 && I % 64 == 0 && J % 64 == 0, "uia_wgrad: I=%d and J=%d must be multiples of 64 (M=%d)", I, J, M);
    const int esz = dtype == UIA_BF16 ? 2 : 4;
    UIA_CHECK_ARG((lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0, "uia_wgrad: alignment");
    UIA_CHECK_ARG(lda >= I && ldb >= J, "uia_wgrad: leading dimension too small");
    UIA_CHECK_ARG(i_valid > 0 && i_valid <= I && j_valid > 0 && j_valid <= J && ldw >= j_valid, "uia_wgrad: valid extent %d x %d (ldw %ld) outside the padded %d x %d", i_valid, j_valid, ldw, I, J);
    WgradGroups gp = {};
    for (int g = 0; g < n; ++g) {
        UIA_CHECK_ARG(A[g] && B[g] && dW[g] && (uintptr_t)A[g] % 16 == 0 && (uintptr_t)B[g] % 16 == 0, "uia_wgrad: null or misaligned tensor (problem %d)", g);
        gp.A[g] = (const bf16_t*)A[g]; gp.B[g] = (const bf16_t*)B[g]; gp.dW[g] = dW[g]; gp.dbias[g] = dbias ? dbias[g] : nullptr; gp.seed[g] = drop_seed ? drop_seed[g] : 0;
    }
    const int chunks = (M + SLAB * CHUNK_SLABS - 1) / (SLAB * CHUNK_SLABS);
    const dim3 grid_f32((I / 64) * (J / 64), chunks);
    const dim3 grid(8 * ((chunks + 7) / 8) * (I / 64) * (J / 64), n);       // bf16 kernel: x = chunks dealt to XCDs (see the kernel), y = problem
    const int lds = 4 * 4096 * 4;   // fp32 path: reduction buffer for four waves at once
    const int lds_bf16 = 2 * SLAB * 128;   // bf16 path: two 16 KiB slabs, re-used as a two-wave reduction buffer
    static UiaDevOnce once_bf16, once_f32;
    UIA_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "uia_wgrad: drop_p=%f outside [0, 1)", (double)drop_p);
    UIA_CHECK_ARG(drop_p == 0.f || (dtype == UIA_BF16 && drop_ld >= drop_col0 + J && drop_col0 >= 0 && drop_col0 % 8 == 0 && drop_ld % 8 == 0 && (size_t)M * (size_t)drop_ld / 8 <= 0xFFFFFFFFull),
                  "uia_wgrad: dropout on the B operand needs bf16, a window of J columns at a multiple of 8 inside rows of drop_ld (a multiple of 8) elements, M*drop_ld <= 2^35");
    UIA_ENSURE_LDS_ATTR(once_bf16, wgrad_bf16_kernel<false>, lds_bf16);
    UIA_ENSURE_LDS_ATTR(once_f32, wgrad_f32_kernel, lds);
    if (dtype == UIA_BF16)
        if (drop_p > 0.f) {
            static UiaDevOnce once_drop;
            UIA_ENSURE_LDS_ATTR(once_drop, wgrad_bf16_kernel<true>, lds_bf16);
            hipLaunchKernelGGL(wgrad_bf16_kernel<true>, grid, dim3(256), lds_bf16, stream, M, I, J, gp, lda, ldb, alpha, ldw, i_valid, j_valid, drop_p, drop_ld, drop_col0);
        } else
        hipLaunchKernelGGL(wgrad_bf16_kernel<false>, grid, dim3(256), lds_bf16, stream, M, I, J, gp, lda, ldb, alpha, ldw, i_valid, j_valid, 0.f, 0l, 0);
    else
        hipLaunchKernelGGL(wgrad_f32_kernel, grid_f32, dim3(256), lds, stream, M, I, J, (const float*)A[0], lda, (const float*)B[0], ldb, alpha, dW[0], dbias ? dbias[0] : nullptr, ldw, i_valid, j_valid);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_wgrad_launch(hipStream_t stream, int dtype, int M, int I, int J, const void* A, long lda, const void* B, long ldb, float alpha,
                     float* dW, float* dbias, long ldw, int i_valid, int j_valid, float drop_p, uint64_t drop_seed, long drop_ld, int drop_col0) {
    UIA_CHECK_ARG(A && B && dW, "uia_wgrad: null tensor");
    return uia_wgrad_group_launch(stream, dtype, 1, M, I, J, &A, lda, &B, ldb, alpha, &dW, &dbias, ldw, i_valid, j_valid, drop_p, &drop_seed, drop_ld, drop_col0);
}
