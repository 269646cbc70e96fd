// gemm.hip — the one dense-contraction kernel of the library:
//     C[M,N] = epilogue( A[M,K] · W[N,K]^T )            ("TN": both operands K-contiguous)
//
// Every Linear of the reference's hot path is an instance of it:
//   forward   y = x·Wᵀ + b         W is the nn.Linear weight [out,in] as stored
//                                  (timm Attention.qkv / proj, Mlp.fc1 / fc2; OpenAI CLIP
//                                  in_proj / out_proj / c_fc / c_proj, src/third_party/openai_clip/model.py:181-188;
//                                  Mona project1 / project2, src/adapters/mona.py:331,358;
//                                  LoRA rank factors, src/adapters/lora.py:78-90)
//   dgrad     dx = dy·W            run as TN against the host-cached transpose Wᵀ [in,out]
//                                  (frozen weights: transposed once at setup; adapter weights: per step)
//
// gfx950 design:
//   * BM×BN output tile per workgroup, 64-lane waves in a WAVES_M×WAVES_N grid, each wave owns
//     (BM/WAVES_M)×(BN/WAVES_N) of C as 16×16 MFMA tiles (v_mfma_f32_16x16x32_bf16, or
//     v_mfma_f32_16x16x4_f32 for the fp32-parity mode: gfx950 has no xf32).
//   * one K-step = 128 bytes of every row (64 bf16 / 32 fp32), staged HBM→LDS with
//     global_load_lds_dwordx4 (no VGPR round trip), double-buffered, one barrier per K-step.
//   * LDS image is lane-linear (a glds requirement); bank conflicts are removed by an XOR
//     swizzle applied to the per-lane SOURCE chunk and again on the ds_read_b128 address.
//   * MFMA is issued "swapped" (W fragment as the A operand) with the W rows of a wave tile
//     permuted so that each lane ends up owning 4·NT consecutive output columns of one row:
//     the epilogue then loads bias/residual and stores C in 16-byte pieces.
//   * workgroup ids are remapped so that each XCD (private L2) walks a contiguous run of tiles.
#include "gemm_epilogue.h"

int uia_gemm_quad_launch(hipStream_t stream, const UiaGemmParams& p, bool specialise, int xflags);   // gemm_quad.hip: tile cfgs 25 / 26
int uia_gemm_quadv_launch(hipStream_t stream, const UiaGemmParams& p, bool specialise, int xflags, int depth);   // gemm_quadv.hip: tile cfgs 27 / 28

namespace {

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_tn_kernel(const UiaGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF_BYTES = A_BYTES + W_BYTES;
    constexpr int A_PER_WAVE = (BM / 8) / NW, W_PER_WAVE = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "staging must divide over the waves");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && (NT % 2) == 0, "wave tile shape");
    constexpr int ESZ = (int)sizeof(T);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- XCD-aware (bijective) workgroup → tile map: blocks b and b+8 share an XCD, so give
    //      each XCD one contiguous run of tiles; inside a run consecutive tiles share the A panel.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane staging sources (K offset added per step). Rows past M / N are clamped:
    //      they feed accumulators that are never stored.
    const char* srcA[A_PER_WAVE];
    const char* srcW[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = m0 + r;
        gm = gm < p.M ? gm : p.M - 1;
        srcA[i] = (const char*)p.A + ((size_t)gm * (size_t)p.lda) * ESZ + c * 16;
    }
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int rl = r & (WTN - 1);
        const int x = ((rl / (4 * NT)) << 1) | ((rl & 3) >> 1);
        const int c = (lane & 7) ^ x;
        int gn = n0 + r;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = (const char*)p.W + ((size_t)gn * (size_t)p.ldw) * ESZ + c * 16;
    }

    // ---- per-lane fragment read offsets inside one LDS buffer
    const int li = lane & 15, g = lane >> 4;
    // A tile (MFMA B operand): row = wm*WTM + 16*mt + li, chunk = g + 4*kk, swizzle (row>>1)&7
    const int offA0 = (wm * WTM + li) * 128 + ((g ^ (li >> 1)) << 4);
    // W tile (MFMA A operand): wave-local row = (li>>2)*4NT + 4*j + (li&3)
    const int xw = ((li >> 2) << 1) | ((li & 3) >> 1);
    const int offW0 = A_BYTES + (wn * WTN + (li >> 2) * (4 * NT) + (li & 3)) * 128 + ((g ^ xw) << 4);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / (128 / ESZ);

    auto stage = [&](int s, int buf) {
        char* base = smem + buf * BUF_BYTES;
        const size_t koff = (size_t)s * 128;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) glds16(srcA[i] + koff, base + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) glds16(srcW[i] + koff, base + A_BYTES + (wave + NW * i) * 1024);
    };

    stage(0, 0);
    for (int s = 0; s < nk; ++s) {
        // tile s has landed (own glds: vmcnt(0); everyone else's: the barrier), and every wave
        // has finished reading the other buffer (its MFMAs of step s-1 precede the barrier).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nk) stage(s + 1, (s + 1) & 1);
        const char* buf = smem + (s & 1) * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[MT], wf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *(const uint4*)(buf + ((offW0 + j * 4 * 128) ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *(const uint4*)(buf + ((offA0 + i * 16 * 128) ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(wf[j], af[i], acc[i][j]);
        }
    }

    gemm_epilogue<T, MT, NT, WTM, WTN>(p, acc, m0, n0, wm, wn, li, g);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(hipStream_t stream, const UiaGemmParams& p) {
    constexpr int LDS = 2 * (BM + BN) * 128;
    auto kern = gemm_tn_kernel<T, BM, BN, WAVES_M, WAVES_N>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, LDS);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES_M * WAVES_N), LDS, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Ping-pong variant (tile cfg 6/7): the workgroup's waves form two groups (wave < NW/2, wave ≥ NW/2), one
// wave of each group per SIMD.  Group 1 runs ONE barrier slot behind group 0, so in every slot one group
// issues its MFMA cluster while the other reads its next fragments from LDS: the matrix pipe sees
// back-to-back clusters instead of [LDS phase | MFMA phase] lockstep.  Per K-tile each wave runs
//     LOAD(kk=0) | COMPUTE(kk=0) | LOAD(kk=1) | COMPUTE(kk=1)        (one raw s_barrier after each)
// Staging: at wall-clock slot 4s every wave issues its global_load_lds for tile s+1 (group 0 at the top of
// LOAD(s,0), group 1 at the top of COMPUTE(s-1,1)); every wave retires them with vmcnt(0) at the end of
// wall-clock slot 4s+3, so the loads stay in flight across four barrier slots.  Hazards:
//   RAW  tile s+1 is first read in slot 4s+4, after the barrier that follows every wave's vmcnt(0);
//   WAR  buffer (s+1)&1 held tile s-1, last read in slot 4s-1 by group 1 whose ds_reads were retired
//        (lgkmcnt(0)) before that slot's barrier.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_tn_pp_kernel(const UiaGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF_BYTES = A_BYTES + W_BYTES;
    constexpr int A_PER_WAVE = (BM / 8) / NW, W_PER_WAVE = (BN / 8) / NW;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && NW % 2 == 0 && (NT % 2) == 0, "tile shape");
    constexpr int ESZ = (int)sizeof(T);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >= NW / 2 ? 1 : 0;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const char* srcA[A_PER_WAVE];
    const char* srcW[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = m0 + r;
        gm = gm < p.M ? gm : p.M - 1;
        srcA[i] = (const char*)p.A + ((size_t)gm * (size_t)p.lda) * ESZ + c * 16;
    }
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int rl = r & (WTN - 1);
        const int x = ((rl / (4 * NT)) << 1) | ((rl & 3) >> 1);
        const int c = (lane & 7) ^ x;
        int gn = n0 + r;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = (const char*)p.W + ((size_t)gn * (size_t)p.ldw) * ESZ + c * 16;
    }
    const int li = lane & 15, g = lane >> 4;
    const int offA0 = (wm * WTM + li) * 128 + ((g ^ (li >> 1)) << 4);
    const int xw = ((li >> 2) << 1) | ((li & 3) >> 1);
    const int offW0 = A_BYTES + (wn * WTN + (li >> 2) * (4 * NT) + (li & 3)) * 128 + ((g ^ xw) << 4);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / (128 / ESZ);
    auto stage = [&](int s) {
        char* base = smem + (s & 1) * BUF_BYTES;
        const size_t koff = (size_t)s * 128;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) glds16_asm(srcA[i] + koff, base + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) glds16_asm(srcW[i] + koff, base + A_BYTES + (wave + NW * i) * 1024);
    };
    uint4 af[MT], wf[NT];
    auto load_frags = [&](const char* buf, int kk) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *(const uint4*)(buf + ((offW0 + j * 4 * 128) ^ (kk << 6)));
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const uint4*)(buf + ((offA0 + i * 16 * 128) ^ (kk << 6)));
    };
    auto compute = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
#define UIA_SLOT_END()                                         \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

#ifdef UIA_GEMM_STAMPS
    unsigned long long t_start = __builtin_amdgcn_s_memtime(), t_pro = 0, t_loop = 0;
#endif
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // tile 0 resident
    __builtin_amdgcn_sched_barrier(0);
#ifdef UIA_GEMM_STAMPS
    t_pro = __builtin_amdgcn_s_memtime();
#endif
    if (grp == 1) {                               // wall-clock slot 0 from group 1's side: prefetch tile 1, then fall one slot behind
        if (nk > 1) stage(1);
        UIA_SLOT_END();
    }
#ifdef UIA_GEMM_STAMPS
    unsigned long long sl[4] = {0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime(), tq;
#define UIA_SLOT_STAMP(i) do { tq = __builtin_amdgcn_s_memtime(); sl[i] += tq - tp; tp = tq; } while (0)
#else
#define UIA_SLOT_STAMP(i)
#endif
    for (int s = 0; s < nk; ++s) {
        const char* buf = smem + (s & 1) * BUF_BYTES;
        // ---- LOAD(s,0)
        if (grp == 0 && s + 1 < nk) stage(s + 1);
        load_frags(buf, 0);
        UIA_SLOT_END();
        UIA_SLOT_STAMP(0);
        // ---- COMPUTE(s,0)
        compute();
        UIA_SLOT_END();
        UIA_SLOT_STAMP(1);
        // ---- LOAD(s,1)
        load_frags(buf, 1);
        if (grp == 1 && s + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // wall-clock slot 4s+3
        UIA_SLOT_END();
        UIA_SLOT_STAMP(2);
        // ---- COMPUTE(s,1)
        if (grp == 1 && s + 2 < nk) stage(s + 2);                                      // wall-clock slot 4(s+1)
        compute();
        if (grp == 0 && s + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // wall-clock slot 4s+3
        UIA_SLOT_END();
        UIA_SLOT_STAMP(3);
    }
#ifdef UIA_GEMM_STAMPS
    if (lane == 0 && uia_stamp_buf && wave == 0) {
        unsigned long long* o = uia_stamp_buf + (size_t)gridDim.x * NW * 4 + (size_t)blockIdx.x * 4;
        o[0] = sl[0]; o[1] = sl[1]; o[2] = sl[2]; o[3] = sl[3];
    }
#endif
    if (grp == 0) {                               // balance group 1's extra slot
        UIA_SLOT_END();
    }
#undef UIA_SLOT_END
#ifdef UIA_GEMM_STAMPS
    t_loop = __builtin_amdgcn_s_memtime();
#endif
    gemm_epilogue_lds<T, MT, NT, WTM, WTN>(p, acc, smem, wave, lane, m0, n0, wm, wn);
#ifdef UIA_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && uia_stamp_buf) {
        unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long* o = uia_stamp_buf + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = t_end;
    }
#endif
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_pp(hipStream_t stream, const UiaGemmParams& p) {
    constexpr int EPI = WAVES_M * WAVES_N * EpiPatch<BM / WAVES_M / 16, BN / WAVES_N>::BYTES_PER_WAVE;
    constexpr int LDS = 2 * (BM + BN) * 128 > EPI ? 2 * (BM + BN) * 128 : EPI;
    auto kern = gemm_tn_pp_kernel<T, BM, BN, WAVES_M, WAVES_N>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, LDS);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WAVES_M * WAVES_N), LDS, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Ring variant (tile cfg 8/9): ping-pong schedule as above, but the staging loads form a CONTINUOUS stream.
// The K dimension is cut into sub-tiles of BKB bytes per row (64 or 128); NBUF sub-tile buffers form a ring in
// LDS and PD = NBUF-1 sub-tiles are kept in flight behind a COUNTED s_waitcnt vmcnt((PD-1)·GPT): the loop never
// drains its loads.  (Measured on the 2-buffer kernel: 15 B/clk/CU from L2 = one 64 KB burst in flight for ~half
// the time at ~2000 cycles latency; the ring keeps 64-96 KB in flight all the time.)
//   step u = one (LOAD, COMPUTE) slot pair = 64 bytes of K per row; sub-tile t = u / SPT, SPT = BKB/64.
//   LOAD(u) of sub-tile t  : the wave issues its share of sub-tile t+PD (1/SPT of its pieces per LOAD slot), so in every
//                            wall slot the loading group's DMA pieces run beside the other group's MFMA cluster
//                            (measured on the 2-buffer kernel: 64 pieces issued in ONE slot made that slot 2444 cycles
//                            against 600-730 for the other three)
//   wall slot 2·t·SPT - 1  : every wave retires sub-tile t     (group 0: end of COMPUTE;     group 1: end of LOAD)
//   RAW: sub-tile t is first read in wall slot 2·t·SPT, after that barrier.   WAR: buffer (t+PD)%NBUF last held
//   sub-tile t-1, whose final ds_reads (group 1, wall slot 2·t·SPT-1) were retired before that slot's barrier.
// 64-byte rows use the 4-entry swizzle table {0,3,2,1} indexed by (row>>2)&3 (A) / the 16-row block of the
// permuted W rows: conflict-free ds_read_b128 for both fragment patterns.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NBUF, int EPI, int LOOP = 0, bool SK = false, bool A2X = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void gemm_tn_ring_kernel(const UiaGemmParams p, const int xflags, const int sk_info) {   // 2 waves per SIMD: one 8-wave
                                                                                                                            // workgroup, or two 4-wave ones, per CU
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_BYTES = BM * BKB, W_BYTES = BN * BKB, BUF_BYTES = A_BYTES + W_BYTES;
    constexpr int RPI = 1024 / BKB;                              // rows per 1 KiB wave-instruction (8 or 16)
    constexpr int CPR = BKB / 16;                                // 16-byte chunks per row (8 or 4)
    constexpr int A_PER_WAVE = (BM / RPI) / NW, W_PER_WAVE = (BN / RPI) / NW;
    constexpr int GPT = A_PER_WAVE + W_PER_WAVE;                 // glds per sub-tile per wave
    constexpr int SPT = BKB / 64;                                // (LOAD, COMPUTE) steps per sub-tile
    constexpr int PD = NBUF - 1;                                 // sub-tiles in flight
    static_assert((BM / RPI) % NW == 0 && (BN / RPI) % NW == 0 && NW % 2 == 0 && NT == 4, "tile shape");
    static_assert(BKB == 64 || BKB == 128, "sub-tile rows are 64 or 128 bytes");
    constexpr int ESZ = (int)sizeof(T);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >= NW / 2 ? 1 : 0;

#if defined(UIA_GEMM_STAMPS) || defined(UIA_GEMM_EXP)
    {   // experiment: de-phase the two workgroups that share a CU (first generation only; later ones inherit the phase)
        const int stg = (xflags >> 10) & 63;                  // delay in units of 4096 cycles
        const bool second = ((xflags >> 16) & 1) ? ((blockIdx.x >> 3) & 1) : ((blockIdx.x >> 8) & 1);
        if (stg && blockIdx.x < (((xflags >> 17) & 1) ? 256 : 512) && second) {   // bit 17: one workgroup per CU (cfg 8): de-phase CUs, not co-residents
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)stg * 4096ull) __builtin_amdgcn_s_sleep(32);
        }
    }
#endif
    const int tiles_n = (p.N + BN - 1) / BN;
    // SPLIT K (sk_info = slices << 2 | phase; the M tail of a launch: a few dozen tiles whose cost is the latency of their K chain, not work).
    //   phase 1: the grid holds `slices` workgroups per tile; each runs its share of the sub-tiles and adds its raw accumulators into the tile's
    //            image in p.splitk_ws (hardware fp32 atomics; the image is zero between uses), no epilogue;
    //   phase 2: one workgroup per tile, no K loop: the accumulators are that image (zeroed again behind the read), then the epilogue as usual.
    // Each output element is still ONE fp32 sum over K; the order in which the slice partials meet is not fixed (results vary in the last bit
    // from run to run: tails only, opt-out ops.TAIL_SPLIT_K).
    // A compile-time variant (SK; instantiated for the tail config with the run-time epilogue only): with the branches in every instantiation the
    // step's main launches lost 1 ms (46.6 vs 45.5 ms, same box, alternating runs).
    const int sk_phase = SK ? (sk_info & 3) : 0, sk_n = SK ? (sk_info >> 2) : 1;
    int nwg = gridDim.x;
    int bid = blockIdx.x, ks = 0;
    if (SK && sk_phase == 1) { ks = bid % sk_n; bid /= sk_n; nwg /= sk_n; }
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_id = bid;
    // Tile order inside an XCD's run (xflags & 255 = GM): GM == 0 walks all column panels of one row panel, then the next row
    // panel; GM > 0 walks GROUPS of GM row panels column by column, rows fastest, so that the ~32 tiles an XCD has in flight
    // form a GM x (32/GM) block: (GM + 32/GM) operand panels per K slice instead of (32/tiles_n + tiles_n), and consecutive
    // waves of a group keep its A panels in the XCD's L2.
    int tm, tn;
    {
        const int gmv = xflags & 255;
        if (gmv > 0) {
            const int tiles_m = (p.M + BM - 1) / BM;
            const int per_group = gmv * tiles_n;
            const int grp_id = bid / per_group, first = grp_id * gmv;
            const int gsz = tiles_m - first < gmv ? tiles_m - first : gmv;
            const int r = bid - grp_id * per_group;
            tn = r / gsz;
            tm = first + (r - tn * gsz);
        } else {
            tm = bid / tiles_n;
            tn = bid - tm * tiles_n;
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    // Folded LayerNorm (EPI_LNFOLD): the (Σ, Σ²) of the wave's rows are requested here, ahead of the K loop, and turned into
    // (rstd, -mean·rstd) in the epilogue: the request is older than every LDS-DMA piece, so the counted waits of the ring are unchanged.
    constexpr bool LNROW = EPI == EPI_GENERIC || (EPI & (EPI_LNFOLD | EPI_RESID_LN)) != 0;      // launch_ring_epi adds the strips to the LDS size for these
    constexpr int LNR = (BM / WAVES_M + 63) / 64;
    float2 lnpre[LNR];
#pragma unroll
    for (int i = 0; i < LNR; ++i) lnpre[i] = float2{0.f, 1.f};
    if (LNROW && p.lnfold_sums) {
#pragma unroll
        for (int i = 0; i < LNR; ++i) {
            const int r = (tid & 63) + 64 * i, m = m0 + (__builtin_amdgcn_readfirstlane(tid >> 6) / WAVES_N) * (BM / WAVES_M) + r;
            if (r < BM / WAVES_M && m < p.M) lnpre[i] = rowsum_load(p.lnfold_sums, (size_t)m);
        }
    } else if (LNROW && p.resid_ln_stats && p.resid_mod == 0 && p.out_group == 0) {
        // deferred LayerNorm of the residual (EPI_RESID_LN): the rows' statistics — (Σ, Σ²) a producing GEMM left, or (mean, rstd) —
        // requested here like the fold's, so that the row passes of the epilogue do not wait on one more load each
#pragma unroll
        for (int i = 0; i < LNR; ++i) {
            const int r = (tid & 63) + 64 * i, m = m0 + (__builtin_amdgcn_readfirstlane(tid >> 6) / WAVES_N) * (BM / WAVES_M) + r;
            if (r < BM / WAVES_M && m < p.M)
                lnpre[i] = p.resid_ln_dim > 0 ? rowsum_load(p.resid_ln_stats, (size_t)m) : *(const float2*)((const float*)p.resid_ln_stats + 2 * (size_t)m);
        }
    }
    // (Round 3, negative: "touching" the epilogue's residual / aux_in tile ahead of the K loop — one dword per 128-byte line, so that the
    // lines travel HBM -> Infinity Cache while the chip is in its K loops — made every fp32-residual launch 6-14 % SLOWER: vmcnt retires in
    // order, so the ring's first counted wait absorbed the 64 MB burst of 256 CUs touching at once (prologue 3 K -> 10-20 K cycles), and the
    // lockstep epilogue it was meant to relieve got only 4-15 % shorter: that phase is bound by its stores.)
    // W may arrive K-BLOCKED ([K·ESZ/64][N][64 bytes], packed once per weight by the host): a sub-tile of a column panel is then
    // one contiguous 16 KiB run, each 1 KiB LDS-DMA piece reads 8 whole 128-byte lines instead of 16 half lines, and the DMA-only
    // K step drops from 3150 to 2880 cycles (N = 2304) / 2350 to 1980 (N = 768): +2…9 % on the whole kernel (profiles/r02_a).
#if defined(UIA_GEMM_STAMPS) || defined(UIA_GEMM_EXP)
    const bool kbA = p.a_kb_rows != 0 || ((xflags >> 8) & 1);         // bit 8, diagnostic only: A addressed as if it were K-blocked
    const bool kbW = p.w_kblocked != 0 || ((xflags >> 9) & 1);
#else
    const bool kbA = p.a_kb_rows != 0;
    const bool kbW = p.w_kblocked != 0;
#endif

    // swizzle of the 16-byte chunk index: 128-byte rows: (row>>1)&7 / 2a|(b>>1) as in the 2-buffer kernels;
    // 64-byte rows: table {0,3,2,1}[(row>>2)&3] for A, [(row_local>>4)&3] for the permuted W rows.
    auto swzA = [](int r) -> int { return BKB == 128 ? ((r >> 1) & 7) : ((0x1230 >> (4 * ((r >> 2) & 3))) & 3); };
    auto swzW = [](int rl) -> int { return BKB == 128 ? ((((rl >> 4) & 3) << 1) | ((rl & 3) >> 1)) : ((0x1230 >> (4 * ((rl >> 4) & 3))) & 3); };

    const char* srcA[A_PER_WAVE];
    const char* srcW[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int r = RPI * (wave + NW * i) + lane / CPR;
        const int c = (lane % CPR) ^ swzA(r);
        int gm = m0 + r;
        gm = gm < p.M ? gm : p.M - 1;
        srcA[i] = kbA ? (const char*)p.A + (size_t)gm * BKB + c * 16 : (const char*)p.A + ((size_t)gm * (size_t)p.lda) * ESZ + c * 16;
    }
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        const int r = RPI * (wave + NW * i) + lane / CPR;
        const int c = (lane % CPR) ^ swzW(r & (WTN - 1));
        int gn = n0 + r;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = kbW ? (const char*)p.W + (size_t)gn * BKB + c * 16 : (const char*)p.W + ((size_t)gn * (size_t)p.ldw) * ESZ + c * 16;
    }
    const size_t kstepA = kbA ? (size_t)(p.a_kb_rows ? p.a_kb_rows : p.M) * BKB : (size_t)BKB, kstepW = kbW ? (size_t)p.N * BKB : (size_t)BKB;
    // K EXTENSION (A2X; the LoRA rank update inside the frozen GEMM, lora.py:87): the last K2 columns of the K loop take their A rows from a second
    // operand — t = drop(x)·Aᵀ, [M, K2] row-major, one per group of a2_group_cols output columns (q | k | v of a fused projection) — while W holds
    // [W | s·B] over the whole K.  A compile-time variant: the step's other launches keep the kernel they had.
    const char* srcA2[A2X ? A_PER_WAVE : 1];
    int ntl1 = 0;
    if constexpr (A2X) {
        ntl1 = ((p.K - p.K2) * ESZ) / BKB;
        const char* a2 = (const char*)p.A2 + (size_t)(p.a2_group_cols > 0 ? n0 / p.a2_group_cols : 0) * (size_t)p.a2_group_stride * ESZ;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) {
            const int r = RPI * (wave + NW * i) + lane / CPR;
            const int c = (lane % CPR) ^ swzA(r);
            int gm = m0 + r;
            gm = gm < p.M ? gm : p.M - 1;
            srcA2[i] = a2 + ((size_t)gm * (size_t)p.lda2) * ESZ + c * 16;
        }
    }
    const int li = lane & 15, g = lane >> 4;
    const int rowA = wm * WTM + li;                                         // + 16·mt  (keeps (row>>1)&7 and (row>>2)&3)
    const int rowW = wn * WTN + (li >> 2) * 16 + (li & 3);                  // + 4·j
    const int offA0 = rowA * BKB + ((g ^ swzA(li)) << 4);
    const int offW0 = A_BYTES + rowW * BKB + ((g ^ swzW((li >> 2) * 16 + (li & 3))) << 4);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntl_all = (p.K * ESZ) / BKB;                                   // sub-tiles
    int t0 = 0, ntl = ntl_all;
    if (!SK) {
    } else if (sk_phase == 1) {
        const int per = (ntl_all + sk_n - 1) / sk_n;
        t0 = ks * per;
        ntl = ntl_all - t0 < per ? ntl_all - t0 : per;
        ntl = ntl < 0 ? 0 : ntl;
    } else if (sk_phase == 2) {
        ntl = 0;
    }
    // part = which 1/SPT of this wave's pieces of sub-tile t (the DMA issue is spread over the wave's LOAD slots)
    auto stage = [&](int t, int part) {
        char* base = smem + (t % NBUF) * BUF_BYTES;
        const size_t koffA = (size_t)(t + t0) * kstepA, koffW = (size_t)(t + t0) * kstepW;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            if (i * SPT / GPT != part && SPT > 1) continue;
            if (i < A_PER_WAVE) {
                if (A2X && t >= ntl1) glds16_asm(srcA2[i] + (size_t)(t - ntl1) * BKB, base + (wave + NW * i) * 1024);
                else glds16_asm(srcA[i] + koffA, base + (wave + NW * i) * 1024);
            } else glds16_asm(srcW[i - A_PER_WAVE] + koffW, base + A_BYTES + (wave + NW * (i - A_PER_WAVE)) * 1024);
        }
    };
    uint4 af[MT], wf[NT];
    auto load_frags = [&](const char* buf, int kk) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *(const uint4*)(buf + ((offW0 + j * 4 * BKB) ^ (kk << 6)));
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const uint4*)(buf + ((offA0 + i * 16 * BKB) ^ (kk << 6)));
    };
    auto compute = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    // retire sub-tile t: everything this wave issued except the (PD-1) newer sub-tiles (fewer near the tail → drain)
    auto retire = [&](int t) {
        if (t + PD - 1 < ntl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * GPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#define UIA_SLOT_END()                                         \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

#ifdef UIA_GEMM_STAMPS
    unsigned long long t_start = __builtin_amdgcn_s_memtime(), t_pro = 0, t_loop = 0;
    const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();      // 100 MHz: in-kernel clock = Δs_memtime / Δs_memrealtime x 100 MHz
#endif
    if constexpr (LOOP >= 1) {
        // ---- FREE-RUNNING loop (tile cfg 15): no LOAD / COMPUTE alternation between two wave groups.  Every wave keeps TWO fragment
        // sets in registers: during step t it issues the LDS-DMA of sub-tile t+NBUF, reads the fragments of sub-tile t+1 into the idle
        // set and runs its 32 MFMAs on the set it read one step earlier, so its own ds_reads fly under its own (and its SIMD partner's)
        // MFMAs and the workgroup meets at ONE barrier per sub-tile instead of two (in-kernel stamps of the ping-pong loop: 650 cycles
        // per slot with MFMAs and barriers alone against 512 of MFMA issue).  Reading a sub-tile into registers one step ahead also
        // frees its ring buffer one step earlier: NBUF sub-tiles stay in flight instead of NBUF-1.
        //   step t:  DMA(t+NBUF) -> buffer t%NBUF   | its last reader: the fragment reads of step t-1, retired (lgkmcnt(0)) before that
        //                                             step's closing barrier                                                    (WAR)
        //            read fragments(t+1)            | sub-tile t+1 was retired (counted vmcnt) before the closing barrier of step t-1  (RAW)
        //            MFMAs(t)
        //            wait: own pieces of sub-tile t+2 (all but the younger sub-tiles t+3, t+4), lgkmcnt(0);  barrier
        static_assert(SPT == 1 && (NBUF == 3 || NBUF == 4), "free-running loop: 64-byte sub-tiles, three or four ring buffers");
        constexpr int PDF = NBUF;                                   // sub-tiles issued ahead
        for (int t = 0; t < PDF && t < ntl; ++t) stage(t, 0);
        // sub-tiles 0 and 1 resident before the first reads: leave the younger ones in flight
        {
            const int younger = (ntl > 2 ? (ntl < PDF ? ntl : PDF) - 2 : 0);
            if (younger >= 2 && PDF >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PDF - 2) * GPT) : "memory");
            else if (younger == 1 || (younger >= 1 && PDF == 3)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef UIA_GEMM_STAMPS
        t_pro = __builtin_amdgcn_s_memtime();
#endif
        uint4 af2[MT], wf2[NT];
        auto read_into = [&](uint4 (&a_)[MT], uint4 (&w_)[NT], const char* buf) {
#pragma unroll
            for (int j = 0; j < NT; ++j) w_[j] = *(const uint4*)(buf + offW0 + j * 4 * BKB);
#pragma unroll
            for (int i = 0; i < MT; ++i) a_[i] = *(const uint4*)(buf + offA0 + i * 16 * BKB);
        };
        // LOOP == 2: the second wave group runs HALF A STEP behind the first one (same program on both waves of a SIMD runs in
        // lockstep otherwise: both in their fragment reads, both in their MFMAs).  Its step is [second half of the previous
        // step's MFMAs | DMA + fragment reads into the set that just became free | first half of this step's MFMAs]: while
        // group 0 reads, group 1 multiplies, and the other way round in the middle of the step.
        constexpr bool STAGGER = LOOP == 2;
        auto mma_lo = [&](uint4 (&a_)[MT], uint4 (&w_)[NT]) {
#pragma unroll
            for (int i = 0; i < MT / 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(w_[j], a_[i], acc[i][j]);
        };
        auto mma_hi = [&](uint4 (&a_)[MT], uint4 (&w_)[NT]) {
#pragma unroll
            for (int i = MT / 2; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(w_[j], a_[i], acc[i][j]);
        };
        auto step = [&](int t, uint4 (&ac)[MT], uint4 (&wc)[NT], uint4 (&an)[MT], uint4 (&wn_)[NT]) {
            if (STAGGER && grp == 1 && t > 0 && !UIA_DIAG_NO_MFMA) { mma_hi(an, wn_); __builtin_amdgcn_sched_barrier(0); }
            if (t + PDF < ntl && !UIA_DIAG_NO_DMA) stage(t + PDF, 0);
            if (t + 1 < ntl && !UIA_DIAG_NO_FRAGS) read_into(an, wn_, smem + ((t + 1) % NBUF) * BUF_BYTES);
            if (!UIA_DIAG_NO_MFMA) {
                if (STAGGER && grp == 1) mma_lo(ac, wc);
                else { mma_lo(ac, wc); mma_hi(ac, wc); }
            }
            if (t + 2 < ntl) {                                      // own pieces of sub-tile t+2 landed; t+3 .. t+PDF may stay in flight
                const int rest = ntl - (t + 3);                     // sub-tiles younger than t+2 that exist
                const int fly = rest < 0 ? 0 : (rest > PDF - 2 ? PDF - 2 : rest);
                if (fly >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPT) : "memory");
                else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        read_into(af, wf, smem);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // step 0 refills buffer 0: every wave's reads of it must be back first
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < ntl; t += 2) {
            step(t, af, wf, af2, wf2);
            if (t + 1 < ntl) step(t + 1, af2, wf2, af, wf);
        }
        if (STAGGER && grp == 1 && !UIA_DIAG_NO_MFMA) {             // the deferred half of the last step
            if (ntl & 1) mma_hi(af, wf);
            else mma_hi(af2, wf2);
        }
    } else {
    for (int t = 0; t < PD && t < ntl; ++t)
#pragma unroll
        for (int part = 0; part < SPT; ++part) stage(t, part);
    retire(0);
    __builtin_amdgcn_s_barrier();                 // sub-tile 0 resident
    __builtin_amdgcn_sched_barrier(0);
#ifdef UIA_GEMM_STAMPS
    t_pro = __builtin_amdgcn_s_memtime();
#endif
    if (grp == 1) {                               // fall one slot behind group 0
        UIA_SLOT_END();
    }
    // LOOP == 3 / 4 (round 3): the LAST TAILI row groups of a COMPUTE slot's MFMA cluster (8 / 16 of its 32 MFMAs) are issued BEHIND the
    // slot's closing barrier, at the head of the wave's next LOAD slot.  With MFMAs and barriers alone the loop takes 650 cycles per slot for
    // 512 cycles of MFMA issue: when a group's cluster ends, the matrix pipe idles for the barrier's turnaround before the other group's
    // cluster starts.  The barrier orders LDS traffic (fragment reads against LDS-DMA), not arithmetic on registers, so the tail of one
    // cluster may run beside the head of the other group's: the pipe sees them back to back.
    constexpr int TAILI = LOOP == 3 ? 2 : (LOOP == 4 ? 4 : 0);
    auto compute_head = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT - TAILI; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto compute_tail = [&]() {
#pragma unroll
        for (int i = MT - TAILI; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nu = ntl * SPT;
    for (int u = 0; u < nu; ++u) {
        const int t = u / SPT, kk = u % SPT;
        const char* buf = smem + (t % NBUF) * BUF_BYTES;
        const bool last_of_tile = kk == SPT - 1;
        if (TAILI > 0 && u > 0 && !UIA_DIAG_NO_MFMA) compute_tail();      // the rest of COMPUTE(u-1): registers only, beside the other group's cluster
        // ---- LOAD(u): this group's DMA pieces of sub-tile t+PD ride beside the other group's MFMA cluster
        if (t + PD < ntl && !UIA_DIAG_NO_DMA) stage(t + PD, kk);
        if (!UIA_DIAG_NO_FRAGS) load_frags(buf, kk);
        if (grp == 1 && last_of_tile && t + 1 < ntl) retire(t + 1);
        UIA_SLOT_END();
        // ---- COMPUTE(u)
        if (!UIA_DIAG_NO_MFMA) { if constexpr (TAILI > 0) compute_head(); else compute(); }
        if (grp == 0 && last_of_tile && t + 1 < ntl) retire(t + 1);
        UIA_SLOT_END();
    }
    if (TAILI > 0 && nu > 0 && !UIA_DIAG_NO_MFMA) compute_tail();
    if (grp == 0) {
        UIA_SLOT_END();
    }
    }   // LOOP == 0
#undef UIA_SLOT_END
#ifdef UIA_GEMM_STAMPS
    t_loop = __builtin_amdgcn_s_memtime();
#endif
    if (SK && sk_phase != 0) {
        // The slices of a tile ADD their raw accumulators into ONE tile-sized fp32 image with hardware float atomics ([tile][wave][MT·NT·4][lane]:
        // 256 contiguous bytes per instruction); the second launch reads it and leaves it zeroed for the next use.  (Slices stored side by side and
        // summed by the second launch made that launch 26 us for four workgroups reading 4 x 128 KB each: it gave back what the first one gained.)
        constexpr size_t WAVE_FLOATS = (size_t)MT * NT * 256;
        float* ws = p.splitk_ws + ((size_t)tile_id * NW + wave) * WAVE_FLOATS + lane;
        if (sk_phase == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) unsafeAtomicAdd(ws + ((i * NT + j) * 4 + r) * 64, acc[i][j][r]);
            return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = ws[((i * NT + j) * 4 + r) * 64];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ws[((i * NT + j) * 4 + r) * 64] = 0.f;
    }
    gemm_epilogue_lds<T, MT, NT, WTM, WTN, EPI>(p, acc, smem, wave, lane, m0, n0, wm, wn,
                                                LNROW ? (float*)(smem + NW * EpiPatch<MT, WTN>::BYTES_PER_WAVE + wave * (WTM * 8)) : nullptr,
                                                LNROW ? lnpre : nullptr);
#ifdef UIA_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && uia_stamp_buf) {
        unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long* o = uia_stamp_buf + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = t_end;
        if (wave == 0) { unsigned long long* q = uia_stamp_buf + (size_t)gridDim.x * NW * 4 + (size_t)blockIdx.x * 4; q[0] = __builtin_amdgcn_s_getreg(63492); q[1] = __builtin_amdgcn_s_getreg(63508); q[2] = t_end - t_start; q[3] = __builtin_amdgcn_s_memrealtime() - r_start; }
    }
#endif
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NBUF, int EPI, int LOOP = 0, bool SK = false, bool A2X = false>
int launch_ring_epi(hipStream_t stream, const UiaGemmParams& p, int xflags, int sk_info = 0) {
    constexpr bool LNROW = EPI == EPI_GENERIC || (EPI & (EPI_LNFOLD | EPI_RESID_LN)) != 0;      // + one (rstd, -mean·rstd) / (mean, rstd) pair per tile row per column of waves
    constexpr int EPB = WAVES_M * WAVES_N * EpiPatch<BM / WAVES_M / 16, BN / WAVES_N>::BYTES_PER_WAVE + (LNROW ? WAVES_N * BM * 8 : 0);
    constexpr int LDS = NBUF * (BM + BN) * BKB > EPB ? NBUF * (BM + BN) * BKB : EPB;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = gemm_tn_ring_kernel<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, EPI, LOOP, SK, A2X>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, LDS);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int grid = (SK && (sk_info & 3) == 1) ? tiles * (sk_info >> 2) : tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES_M * WAVES_N), LDS, stream, p, xflags, sk_info);
    UIA_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Persistent ring kernel (tile cfg 12): one workgroup per CU walks its share of the 256×256 tiles.  Same K loop as the
// ring kernel (64-byte sub-tiles, 4-deep ring, two wave groups one barrier slot apart); what changes is the seam
// between tiles: right after the last barrier of a tile's K loop the workgroup issues the LDS-DMA of the NEXT tile's
// first three sub-tiles into ring buffers 0..2 and runs the epilogue out of buffer 3 (16-row swizzled patches), so the
// prologue's HBM/L2 latency (3-6 K cycles per tile in the stamps) and the workgroup hand-over (≈1.5 K) hide under the
// epilogue.  Tile order: the blocks of one XCD (blockIdx & 7) own a contiguous range of tiles and take them
// round-robin, so at any time an XCD's CUs work on neighbouring tiles that share A row panels in its L2.
// vmcnt bookkeeping at the seam: the epilogue's loads/stores are younger than the three prefetched sub-tiles, so the
// counted wait for sub-tile 0 (all but the 2·GPT youngest operations) also waits for them — correct, slightly early.
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm_tn_persist_kernel(const UiaGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, WAVES_M = 2, WAVES_N = 4, BKB = 64, NBUF = 4;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_BYTES = BM * BKB, W_BYTES = BN * BKB, BUF_BYTES = A_BYTES + W_BYTES;
    constexpr int RPI = 1024 / BKB, CPR = BKB / 16;
    constexpr int A_PER_WAVE = (BM / RPI) / NW, W_PER_WAVE = (BN / RPI) / NW;
    constexpr int GPT = A_PER_WAVE + W_PER_WAVE;
    constexpr int PD = NBUF - 1;
    constexpr int ESZ = (int)sizeof(T);
    static_assert(NW * 16 * WTN * 4 <= BUF_BYTES, "epilogue patches must fit one ring buffer");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >= NW / 2 ? 1 : 0;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles = ((p.M + BM - 1) / BM) * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nslots = ((int)gridDim.x - xcd + 7) >> 3;                      // workgroups that share this XCD
    const int tq = tiles >> 3, tr = tiles & 7;
    const int t_lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_hi = t_lo + (xcd < tr ? tq + 1 : tq);
    int tile = t_lo + slot;
    if (tile >= t_hi) return;

    auto swzA = [](int r) -> int { return (0x1230 >> (4 * ((r >> 2) & 3))) & 3; };
    auto swzW = [](int rl) -> int { return (0x1230 >> (4 * ((rl >> 4) & 3))) & 3; };
    const bool kbA = p.a_kb_rows != 0, kbW = p.w_kblocked != 0;
    const char* srcA[A_PER_WAVE];
    const char* srcW[W_PER_WAVE];
    int m0 = 0, n0 = 0;
    auto point_at = [&](int tl) {
        const int tm = tl / tiles_n, tn = tl - tm * tiles_n;
        m0 = tm * BM;
        n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) {
            const int r = RPI * (wave + NW * i) + lane / CPR;
            const int c = (lane % CPR) ^ swzA(r);
            int gm = m0 + r;
            gm = gm < p.M ? gm : p.M - 1;
            srcA[i] = kbA ? (const char*)p.A + (size_t)gm * BKB + c * 16 : (const char*)p.A + ((size_t)gm * (size_t)p.lda) * ESZ + c * 16;
        }
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) {
            const int r = RPI * (wave + NW * i) + lane / CPR;
            const int c = (lane % CPR) ^ swzW(r & (WTN - 1));
            int gn = n0 + r;
            gn = gn < p.N ? gn : p.N - 1;
            srcW[i] = kbW ? (const char*)p.W + (size_t)gn * BKB + c * 16 : (const char*)p.W + ((size_t)gn * (size_t)p.ldw) * ESZ + c * 16;
        }
    };
    // K-blocked operands ([K·ESZ/64][rows][64 B]: uia_gemm_desc.w_kblocked / a_kb_rows): one K step = one plane further
    const size_t kstepA = kbA ? (size_t)p.a_kb_rows * BKB : (size_t)BKB, kstepW = kbW ? (size_t)p.N * BKB : (size_t)BKB;
    const int li = lane & 15, g = lane >> 4;
    const int rowA = wm * WTM + li;
    const int rowW = wn * WTN + (li >> 2) * 16 + (li & 3);
    const int offA0 = rowA * BKB + ((g ^ swzA(li)) << 4);
    const int offW0 = A_BYTES + rowW * BKB + ((g ^ swzW((li >> 2) * 16 + (li & 3))) << 4);
    const int ntl = (p.K * ESZ) / BKB;

    auto stage = [&](int t) {
        char* base = smem + (t % NBUF) * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) glds16_asm(srcA[i] + (size_t)t * kstepA, base + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) glds16_asm(srcW[i] + (size_t)t * kstepW, base + A_BYTES + (wave + NW * i) * 1024);
    };
    uint4 af[MT], wf[NT];
    auto load_frags = [&](const char* buf) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *(const uint4*)(buf + offW0 + j * 4 * BKB);
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const uint4*)(buf + offA0 + i * 16 * BKB);
    };
    f32x4 acc[MT][NT];
    auto compute = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = MfmaTile<T>::mma(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto retire = [&](int t) {
        if (t + PD - 1 < ntl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * GPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#define UIA_SLOT_END()                                         \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

    point_at(tile);
    for (int t = 0; t < PD && t < ntl; ++t) stage(t);
    while (true) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        retire(0);
        __builtin_amdgcn_s_barrier();                 // sub-tile 0 resident; every wave is out of the previous epilogue
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {                               // fall one slot behind group 0
            UIA_SLOT_END();
        }
        for (int t = 0; t < ntl; ++t) {
            const char* buf = smem + (t % NBUF) * BUF_BYTES;
            if (t + PD < ntl) stage(t + PD);
            load_frags(buf);
            if (grp == 1 && t + 1 < ntl) retire(t + 1);
            UIA_SLOT_END();
            compute();
            if (grp == 0 && t + 1 < ntl) retire(t + 1);
            UIA_SLOT_END();
        }
        if (grp == 0) {
            UIA_SLOT_END();
        }
        const int cm0 = m0, cn0 = n0;
        const int next = tile + nslots;
        const bool more = next < t_hi;
        if (more) {                                   // next tile's head rides under this tile's epilogue
            point_at(next);
            for (int t = 0; t < PD && t < ntl; ++t) stage(t);
        }
        constexpr bool LNROW = EPI == EPI_GENERIC || (EPI & EPI_LNFOLD) != 0;      // the per-wave (rstd, -mean·rstd) strips sit behind the ring
        gemm_epilogue_lds<T, MT, NT, WTM, WTN, EPI, true>(p, acc, smem + (NBUF - 1) * BUF_BYTES, wave, lane, cm0, cn0, wm, wn,
                                                          LNROW ? (float*)(smem + NBUF * BUF_BYTES + wave * (WTM * 8)) : nullptr);
        if (!more) break;
        tile = next;
    }
#undef UIA_SLOT_END
}

template <typename T, int EPI>
int launch_persist_epi(hipStream_t stream, const UiaGemmParams& p) {
    constexpr int LDS = 4 * 512 * 64 + ((EPI == EPI_GENERIC || (EPI & EPI_LNFOLD) != 0) ? 4 * 256 * 8 : 0);
    auto kern = gemm_tn_persist_kernel<T, EPI>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, LDS);
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int ncu = uia_num_cus();
    hipLaunchKernelGGL(kern, dim3(tiles < ncu ? tiles : ncu), dim3(512), LDS, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}


template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int BKB, int NBUF, int LOOP = 0>
int launch_ring(hipStream_t stream, const UiaGemmParams& p, bool specialise, int xflags, int sk_info = 0) {
    if constexpr (BM == 128 && BN == 256 && NBUF == 4 && LOOP == 0) {          // tile cfg 13 only
        if (sk_info != 0) return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, EPI_GENERIC, LOOP, true>(stream, p, xflags, sk_info);
    }
    if constexpr (BN == 256 && NBUF == 4 && LOOP == 0 && BKB == 64 && sizeof(T) == 2) {      // tile cfgs 8 and 13: the K-extension variant, three masks + the run-time one
        if (p.K2 > 0) {
            switch (epi_mask_of(p)) {
                case EPI_BIAS | EPI_OUTT: return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, (EPI_BIAS | EPI_OUTT), LOOP, false, true>(stream, p, xflags);
                case EPI_OUTT: return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, (EPI_OUTT), LOOP, false, true>(stream, p, xflags);
                case EPI_BIAS | EPI_RESID | EPI_OUT32: return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, (EPI_BIAS | EPI_RESID | EPI_OUT32), LOOP, false, true>(stream, p, xflags);
                default: return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, EPI_GENERIC, LOOP, false, true>(stream, p, xflags);
            }
        }
    }
    if (specialise) {
        switch (epi_mask_of(p)) {    // the six masks of a training step, by time spent (tools/gemm_census.py)
#define UIA_EPI_CASE(MASK) case (MASK): return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, (MASK), LOOP>(stream, p, xflags)
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUT32);                    // proj / fc2 / Mona project2 forward
            UIA_EPI_CASE(EPI_BIAS | EPI_RESIDT | EPI_OUT32);                   // post-LN (BERT) sub-layer sums on the T residual
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUT32);     // post-LN (BERT) sub-layer sums, LayerNorm of the previous sum applied here
            UIA_EPI_CASE(EPI_OUTT);                                            // dgrads
            UIA_EPI_CASE(EPI_BIAS | EPI_OUTT);                                 // QKV
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_OUTT);                      // fc1, frozen tower
            UIA_EPI_CASE(EPI_DGELU | EPI_OUTT);                                // fc2 dgrad through GELU'
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT);        // fc1 with the pre-activation stashed
            // the same three with QuickGELU (OpenAI CLIP towers; on the run-time epilogue fc1 of ViT-L/14 took 410 us against 275 for fc2's 4x longer K)
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_OUTT);
            UIA_EPI_CASE(EPI_QUICK | EPI_DGELU | EPI_OUTT);
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT);
            // LayerNorm folded into its neighbours (bf16 step): producers write fp32 + T rows and their row sums, consumers normalise the accumulators
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM);                  // proj / fc2 / Mona project2
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM);   // BERT sub-layer sums
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID_LO | EPI_RESID_LN | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM);   // the same on three-byte tensors (round 4): 6 epilogue bytes per element instead of 10
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM);                     // image tower, output projection: x1 leaves as a three-byte tensor
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID_LO | EPI_OUT32);                                           // ... and enters fc2's epilogue as one
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_RESID_LN | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM);      // text tower, first layer: the embedding LayerNorm's fp32 rows in, a three-byte sum out
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID_LO | EPI_RESID_LN | EPI_OUT32);                            // text tower, last layer: a three-byte sum in, fp32 rows out for the pooler
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID_LO | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM);                   // image tower, output projection behind a Mona adapter (round 5): the block's input arrives as a three-byte tensor
            UIA_EPI_CASE(EPI_BIAS | EPI_OUTT | EPI_LNFOLD);                                          // QKV
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD);                               // fc1, frozen tower
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD);                 // fc1 with the pre-activation stashed
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD);                   // fc1 of a frozen OpenAI-CLIP block (CLIPSeg's backbone)
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD);
#undef UIA_EPI_CASE
            default: break;
        }
    }
    return launch_ring_epi<T, BM, BN, WAVES_M, WAVES_N, BKB, NBUF, EPI_GENERIC, LOOP>(stream, p, xflags);
}

template <typename T>
int launch_persist(hipStream_t stream, const UiaGemmParams& p) {
    switch (epi_mask_of(p)) {
#define UIA_EPI_CASE(MASK) case (MASK): return launch_persist_epi<T, (MASK)>(stream, p)
        UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUT32);
        UIA_EPI_CASE(EPI_BIAS | EPI_RESIDT | EPI_OUT32);
        UIA_EPI_CASE(EPI_OUTT);
        UIA_EPI_CASE(EPI_BIAS | EPI_OUTT);
        UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_OUTT);
        UIA_EPI_CASE(EPI_DGELU | EPI_OUTT);
        UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT);
        UIA_EPI_CASE(EPI_BIAS | EPI_OUTT | EPI_LNFOLD);
        UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD);
        UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD);
#undef UIA_EPI_CASE
        default: break;
    }
    return launch_persist_epi<T, EPI_GENERIC>(stream, p);
}

// ------------------------------------------------------------------------------------------------
// Tile cfg 16 — N = 64 (the adapters' down-projections and their data gradients: Mona project1 forward, project2 dgrad; 768 → 64):
// 2·M·64·K FLOP against M·K·2 bytes of A is 64 FLOP per byte, i.e. the launch is a stream over A.  One workgroup per CU keeps ALL of W
// (64 x K bf16, ≤ 160 KB) in LDS for its lifetime; every wave walks 16-row tiles of A with its fragments loaded straight from HBM
// into registers (16 rows x 64 contiguous bytes per instruction, CH instructions in flight per wave) and W as the MFMA A operand,
// so that a lane ends up with four consecutive outputs of one row.  The 256 x 64 tile config had 197 workgroups for 256 CUs, each
// streaming its rows through a double-buffered LDS ring with a barrier per K step: 35 / 34 us inside the step; this one 31 / 26 (14 at HBM rate).
template <int CH, int NWV = 8>      // CH: K steps (32 columns each) whose A fragments a wave requests before it starts multiplying; NWV: waves per workgroup
__global__ __launch_bounds__(64 * NWV) void gemm_skinny64_kernel(const UiaGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, KS = K >> 5;
    const int ldw_b = 2 * K + 16;                          // row stride of the W image: +16 B staggers the rows over the banks
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const int cpr = K >> 3;                            // 16-byte chunks per row of W
        for (int c = tid; c < 64 * cpr; c += 64 * NWV) {
            const int r = c / cpr, cc = c - r * cpr;
            *(uint4*)(smem + r * ldw_b + cc * 16) = *(const uint4*)((const char*)p.W + ((size_t)r * p.ldw) * 2 + cc * 16);
        }
    }
    __syncthreads();
    float bias[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[nt][r] = p.bias ? p.bias[16 * nt + 4 * g + r] : 0.f;
    const char* wfrag = smem + li * ldw_b + g * 16;        // W row 16nt + li, bytes 64ks + 16g: + nt·16·ldw_b + ks·64
    const bool drop_a = p.drop_where == 1;
    const uint32_t drop_th = dropout_thresh16(p.drop_p);
    const float drop_inv = 1.0f / (1.0f - p.drop_p);
    const int ntiles = (p.M + 15) >> 4;
    // tiles are dealt wave-major: wave w of workgroup b starts at tile w·grid + b, so a launch with fewer tiles than wave slots still puts work on
    // every CU (ViT-L/14: 2056 tiles — dealt workgroup-major they filled 129 of the 256 CUs, or left eight tiles for a second pass)
    for (int tile = wave * gridDim.x + blockIdx.x; tile < ntiles; tile += gridDim.x * NWV) {
        const int m = 16 * tile + li;
        const int mc = m < p.M ? m : p.M - 1;
        const char* arow = (const char*)p.A + ((size_t)mc * p.lda) * 2 + g * 16;
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int ks0 = 0; ks0 < KS; ks0 += CH) {
            uint4 a[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = ks0 + i < KS ? *(const uint4*)(arow + (ks0 + i) * 64) : uint4{0u, 0u, 0u, 0u};
            if (drop_a) {                                  // LoRA input dropout on the operand in flight: what uia_dropout would have written, bit for bit
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if (ks0 + i < KS) {
                        const uint32_t keep = dropout_keep8(p.drop_seed, (uint32_t)(((size_t)mc * (size_t)K + (size_t)((ks0 + i) * 32 + g * 8)) >> 3), drop_th);
                        bf16x8 v = __builtin_bit_cast(bf16x8, a[i]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((keep >> e) & 1u ? (float)v[e] * drop_inv : 0.f);
                        a[i] = __builtin_bit_cast(uint4, v);
                        if (p.a_drop_out && m < p.M) *(uint4*)((char*)p.a_drop_out + ((size_t)m * p.lda) * 2 + g * 16 + (ks0 + i) * 64) = a[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (ks0 + i < KS) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const uint4 wf = *(const uint4*)(wfrag + nt * 16 * ldw_b + (ks0 + i) * 64);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, a[i]), acc[nt], 0, 0, 0);
                    }
                }
            }
        }
        if (m < p.M) {                                     // lane: row m, columns 16nt + 4g .. +3
            bf16_t* orow = (bf16_t*)p.outT + (size_t)m * p.ldo + 4 * g;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 v = {acc[nt][0] + bias[nt][0], acc[nt][1] + bias[nt][1], acc[nt][2] + bias[nt][2], acc[nt][3] + bias[nt][3]};
                store4(orow + 16 * nt, v);
            }
        }
    }
}

// cfg 16 takes: bf16, N == 64, K % 32 == 0 with the W image inside the LDS, and nothing in the epilogue but an optional bias and the T output
inline bool skinny64_ok(const UiaGemmParams& p, int esz) {
    return esz == 2 && p.N == 64 && p.K % 32 == 0 && 64 * (2 * p.K + 16) <= 160 * 1024 && p.alpha == 1.0f && p.outT && !p.out32 && !p.act && !p.dact && p.drop_where != 2 &&
           !p.aux_out && !p.resid && !p.residT && p.out_group == 0 && !p.w_kblocked && !p.rowsum_out && !p.lnfold_sums;
}

int launch_skinny64(hipStream_t stream, const UiaGemmParams& p) {
    const int lds = 64 * (2 * p.K + 16);
    static UiaDevOnce once8;
    UIA_ENSURE_LDS_ATTR(once8, gemm_skinny64_kernel<8>, 160 * 1024);
    const int ncu = uia_num_cus();
    const int ntiles = (p.M + 15) / 16;
    const int grid = ntiles < ncu ? ntiles : ncu;
    {   // Sixteen waves per CU when eight would need a second, partly filled pass over the tiles (M = 50 432: 3152 tiles for 2048 waves; ViT-L/14's
        // M = 32 896: 2056): every tile is then in flight at once (4 waves per SIMD at <= 128 VGPRs), the dropout variant included.
        static UiaDevOnce once16;
        const char* env = getenv("UIA_SKINNY_WAVES");
        const int want = env ? atoi(env) : 0;
        if (want == 16 || (want == 0 && ntiles > 8 * ncu)) {
            UIA_ENSURE_LDS_ATTR(once16, (gemm_skinny64_kernel<8, 16>), 160 * 1024);
            hipLaunchKernelGGL((gemm_skinny64_kernel<8, 16>), dim3(grid), dim3(1024), lds, stream, p);
            UIA_CHECK_LAUNCH();
            return 0;
        }
    }
    // eight K steps (8 KiB per wave) in flight: requesting a whole 768-wide row tile at once (24 steps, 160 VGPRs) measured slower
    // inside the step (34 / 30 us against 31 / 26 for the two Mona launches)
    hipLaunchKernelGGL(gemm_skinny64_kernel<8>, dim3(grid), dim3(512), lds, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Tile cfg 23 — K = 64 (the LoRA rank update y += s·t·Bᵀ and its data gradient dx += drop(s·q·A), r padded to 64; lora.py:87): 2·M·N·64 FLOP
// against 4·M·N bytes of result read and written back is 32 FLOP per byte, i.e. the launch is a read-modify-write stream over the result.
// The mirror image of cfg 16: one workgroup per CU keeps ALL of W (N x 64 bf16, N <= 1088) in LDS; a wave's unit of work is 16 rows x 64
// columns — fine enough that M = 128·257 rows (ViT-L/14) deal out evenly, where the 128 x 256 tiles of cfg 14 left 1028 tiles for 512
// slots (61 us per launch for 134 MB).  The A fragment of a unit (16 rows x 64 k) comes straight from global memory (the whole operand is
// 4 MB: L2), W rows are read from LDS in the order that leaves a lane 16 CONSECUTIVE columns of one row (column 64c + 16g + 4nt + r from MFMA
// row 4g + r of tile nt), so residual loads and stores are 32 contiguous bytes per lane and a full 128-byte line per row.  The residual of the
// NEXT unit is requested before the current one is multiplied.  Dropout on the accumulator (drop_where = 2) is drawn per eight columns.
template <bool OUT32>
__global__ __launch_bounds__(512) void gemm_wide64_kernel(const UiaGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDWB = 128 + 16;                         // row stride of the W image
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int c = tid; c < p.N * 8; c += 512) {
        const int r = c >> 3, cc = c & 7;
        *(uint4*)(smem + r * LDWB + cc * 16) = *(const uint4*)((const char*)p.W + ((size_t)r * p.ldw) * 2 + cc * 16);
    }
    __syncthreads();
    const int nch = p.N >> 6, ntiles = (p.M + 15) >> 4;
    const int nunits = ntiles * nch, ustep = gridDim.x * 8;
    const bool drop = p.drop_where == 2;
    const uint32_t drop_th = dropout_thresh16(p.drop_p);
    const float drop_inv = 1.0f / (1.0f - p.drop_p);
    const bf16_t* residT = (const bf16_t*)p.residT;
    bf16_t* outT = (bf16_t*)p.outT;
    // W fragment of (chunk c, tile nt, k-step ks): row 64c + 16(li>>2) + 4nt + (li&3), bytes 64ks + 16g
    const char* wfrag = smem + (16 * (li >> 2) + (li & 3)) * LDWB + g * 16;

    struct Unit { uint4 a0, a1; uint4 rt0, rt1; f32x4 r0, r1, r2, r3; };
    auto request = [&](int u, Unit& q) {
        const int rt = u / nch, c = u - rt * nch;
        const int m = 16 * rt + li, mc = m < p.M ? m : p.M - 1;
        const char* arow = (const char*)p.A + ((size_t)mc * p.lda) * 2 + g * 16;
        q.a0 = *(const uint4*)arow;
        q.a1 = *(const uint4*)(arow + 64);
        const int n = 64 * c + 16 * g;
        if (OUT32) {
            if (p.resid) {
                const float* rr = p.resid + (size_t)mc * p.ldr + n;
                q.r0 = *(const f32x4*)rr; q.r1 = *(const f32x4*)(rr + 4); q.r2 = *(const f32x4*)(rr + 8); q.r3 = *(const f32x4*)(rr + 12);
            }
        } else if (residT) {
            const bf16_t* rr = residT + (size_t)mc * p.ldrT + n;
            q.rt0 = *(const uint4*)rr; q.rt1 = *(const uint4*)(rr + 8);
        }
    };
    auto process = [&](int u, const Unit& q) {
        const int rt = u / nch, c = u - rt * nch;
        const int m = 16 * rt + li;
        const int n = 64 * c + 16 * g;
        const char* wc = wfrag + (size_t)(64 * c) * LDWB;
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const uint4 w0 = *(const uint4*)(wc + (4 * nt) * LDWB), w1 = *(const uint4*)(wc + (4 * nt) * LDWB + 64);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, q.a0), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, q.a1), acc[nt], 0, 0, 0);
        }
        if (m >= p.M) return;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {                   // alpha and bias in one fma, as the run-time epilogue of the tiled kernels has them
            const f32x4 b = p.bias ? *(const f32x4*)(p.bias + n + 4 * nt) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * nt + r] = fmaf(acc[nt][r], p.alpha, b[r]);
        }
        if (drop) {
            const uint32_t grp = (uint32_t)(((size_t)m * (size_t)p.N + (size_t)n) >> 3);
            const uint32_t k0 = dropout_keep8(p.drop_seed, grp, drop_th), k1 = dropout_keep8(p.drop_seed, grp + 1, drop_th);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = (k0 >> e) & 1u ? v[e] * drop_inv : 0.f; v[8 + e] = (k1 >> e) & 1u ? v[8 + e] * drop_inv : 0.f; }
        }
        if (OUT32) {
            if (p.resid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] += q.r0[r]; v[4 + r] += q.r1[r]; v[8 + r] += q.r2[r]; v[12 + r] += q.r3[r]; }
            }
            float* o = p.out32 + (size_t)m * p.ldo32 + n;
#pragma unroll
            for (int h = 0; h < 4; ++h) *(f32x4*)(o + 4 * h) = f32x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
        } else {
            if (residT) {
                const bf16x8 x0 = __builtin_bit_cast(bf16x8, q.rt0), x1 = __builtin_bit_cast(bf16x8, q.rt1);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] += (float)x0[e]; v[8 + e] += (float)x1[e]; }
            }
            bf16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o0[e] = (bf16_t)v[e]; o1[e] = (bf16_t)v[8 + e]; }
            bf16_t* o = outT + (size_t)m * p.ldo + n;
            *(bf16x8*)o = o0;
            *(bf16x8*)(o + 8) = o1;
        }
    };
    // THREE units requested ahead of the one being multiplied (four register sets, the loop unrolled by four): with one unit ahead a wave had
    // 4 KB in flight and the launch was bound by the round trip of a unit (49 us for 134 MB at M = 32 896, N = 1024)
    int u = blockIdx.x * 8 + wave;
    Unit q0, q1, q2, q3;
    if (u < nunits) request(u, q0);
    if (u + ustep < nunits) request(u + ustep, q1);
    if (u + 2 * ustep < nunits) request(u + 2 * ustep, q2);
    while (u < nunits) {
        if (u + 3 * ustep < nunits) request(u + 3 * ustep, q3);
        process(u, q0);
        u += ustep;
        if (u >= nunits) break;
        if (u + 3 * ustep < nunits) request(u + 3 * ustep, q0);
        process(u, q1);
        u += ustep;
        if (u >= nunits) break;
        if (u + 3 * ustep < nunits) request(u + 3 * ustep, q1);
        process(u, q2);
        u += ustep;
        if (u >= nunits) break;
        if (u + 3 * ustep < nunits) request(u + 3 * ustep, q2);
        process(u, q3);
        u += ustep;
    }
}

// cfg 23 takes: bf16, K == 64, N a multiple of 64 whose W image fits the LDS, row-major operands, and an epilogue of alpha, bias, dropout on the
// accumulator and ONE residual of the output's type (T residual -> T output, or fp32 residual -> fp32 output)
inline bool wide64_ok(const UiaGemmParams& p, int esz) {
    const bool t_form = p.outT && !p.out32 && !p.resid, f_form = p.out32 && !p.outT && !p.residT;
    return esz == 2 && p.K == 64 && p.N % 64 == 0 && p.N * (128 + 16) <= 160 * 1024 && (t_form || f_form) && !p.act && !p.dact && !p.aux_out && p.out_group == 0 &&
           p.resid_mod == 0 && !p.w_kblocked && !p.a_kb_rows && !p.outT_kb_rows && !p.rowsum_out && !p.lnfold_sums && !p.resid_ln_stats && p.drop_where != 1 &&
           (!p.outT || p.ldo % 8 == 0) && (!p.residT || p.ldrT % 8 == 0);
}

int launch_wide64(hipStream_t stream, const UiaGemmParams& p) {
    const int lds = p.N * (128 + 16);
    static UiaDevOnce once_t, once_f;
    UIA_ENSURE_LDS_ATTR(once_t, gemm_wide64_kernel<false>, 160 * 1024);
    UIA_ENSURE_LDS_ATTR(once_f, gemm_wide64_kernel<true>, 160 * 1024);
    const int ncu = uia_num_cus();
    const long units = (long)((p.M + 15) / 16) * (p.N / 64);
    int grid = (int)((units + 7) / 8);
    grid = grid < ncu ? grid : ncu;
    if (p.out32) hipLaunchKernelGGL(gemm_wide64_kernel<true>, dim3(grid), dim3(512), lds, stream, p);
    else hipLaunchKernelGGL(gemm_wide64_kernel<false>, dim3(grid), dim3(512), lds, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int launch_typed(hipStream_t stream, const UiaGemmParams& p, int cfg_in) {
    // bits 8.. of the tile argument carry experiment knobs for the ring kernels (tile-order group size, diagnostic layouts);
    // 0 there = the launcher's own choice.
    int cfg = cfg_in & 255, xflags = cfg_in >> 8;
    int sk_info = 0;
#if !defined(UIA_GEMM_EXP) && !defined(UIA_GEMM_STAMPS)
    // bits 16-21: K slices, bits 22-23: split-K phase (1 = partial sums of one slice per workgroup into splitk_ws, 2 = sum of the slices + epilogue)
    {
        const int slices = (cfg_in >> 16) & 63, phase = (cfg_in >> 22) & 3;
        xflags &= 255;
        if (phase != 0) {
            if (p.K2 > 0) {      // the split-K variant of the ring kernel has no K-extension operand: it would read K columns of A and ignore A2
                uia_set_error("uia_gemm: split K (phase %d) cannot be combined with the K extension (A2 / K2 = %d)", phase, p.K2);
                return -1;
            }
            if (phase == 3 || slices < 2 || cfg != 13 || !p.splitk_ws || (uintptr_t)p.splitk_ws % 16 != 0 || p.K * (int)sizeof(T) / 64 < slices) {
                uia_set_error("uia_gemm: split K needs phase 1 or 2, 2..63 slices (at most one per 64 bytes of K), the half-height tail config (tile cfg 13) and a 16-byte aligned splitk_ws; "
                              "got phase %d, %d slices, tile cfg %d", phase, slices, cfg);
                return -1;
            }
            sk_info = (slices << 2) | phase;
        }
    }
#endif
    const int gm_req = xflags & 255;                           // tile-order group: 0 = the launcher's choice, 255 = none (row-panel-major)
    // cfg: 0 = auto. Tile choice is a pure speed knob (results are identical for every config
    // up to fp32 summation order inside a K-step, which does not depend on the tile).
    if (cfg == 0) {
        if (p.N <= 64) cfg = ((p.M > 2048 || p.drop_where == 1) && skinny64_ok(p, (int)sizeof(T))) ? 16
                                 : ((p.M > 2048 && sizeof(T) == 2 && p.K >= 1024) ? 14 : 4);   // a W image too large for the stream kernel's LDS (CLIPSeg's K = 2048
                                                                                                // reductions): the 3-deep ring streams A faster than the 2-buffer tiles (33 vs 42 us)
        else if (p.M <= 2048) cfg = (sizeof(T) == 4 && ((p.M + 127) / 128) * ((p.N + 127) / 128) < 64) ? 21 : 3;
            // fp32 MFMA issues 256 FLOP per clock per CU: the 8-12 workgroups of a [256, 512-768] head projection on 128 x 128 tiles were bound
            // by their own CUs' matrix pipes (69 us at K = 768); 32 x 64 tiles spread the same MFMA sequence per element over 64-96 CUs
        else if (wide64_ok(p, (int)sizeof(T)) && (p.residT || p.resid || p.drop_where == 2) && !p.resid_lo8 && !p.out_lo8) cfg = 23;   // K = 64 read-modify-write stream (LoRA rank update)
        else if (p.K * (int)sizeof(T) <= 128) cfg = 14;   // one K step (Mona project2 / project1-dgrad, K = 64): nothing but prologue + epilogue, HBM-bound:
                                                          // half-height tiles, two workgroups per CU (70.9 vs 86.6 us and 20.0 vs 26.3 us at M = 50 432)
        else cfg = 8;            // 256x256 ping-pong, 4-deep 64-byte ring, LDS-staged epilogue: best measured on every large shape.
                                 // (cfg 12, the persistent variant, is +2-3.5 % on store-only epilogues in isolation, -15-25 % on the
                                 //  fp32-residual ones, and a net loss inside the two-stream training step: opt-in only.)
    }
    if (p.K2 > 0 && (cfg_in & 255) == 0) cfg = p.M <= 2048 ? 13 : 8;       // the K extension lives on the 64-byte-sub-tile ring kernels
    if (p.K2 > 0 && !(cfg == 8 || cfg == 13 || cfg == 25 || cfg == 26)) {
        uia_set_error("uia_gemm: the K extension (A2 / K2) runs on tile cfgs 8 and 13, not %d", cfg);
        return -1;
    }
    if (p.drop_where == 1 && cfg != 16) {
        uia_set_error("uia_gemm: dropout on the A operand (drop_where = 1) is the N = 64 stream kernel's (tile cfg 16: bf16, N == 64, M > 2048, bias + T output only), not tile cfg %d", cfg);
        return -1;
    }
    const bool ring = cfg == 8 || cfg == 9 || cfg == 10 || cfg == 12 || cfg == 13 || cfg == 14 || cfg == 15 || (cfg >= 17 && cfg <= 20) || cfg == 24 || cfg == 25 || cfg == 26 || cfg == 27 || cfg == 28 || cfg == 29;
    if ((p.a_kb_rows || p.outT_kb_rows) && !(cfg == 8 || cfg == 10 || cfg == 12 || cfg == 13 || cfg == 14 || cfg == 15 || (cfg >= 17 && cfg <= 20) || cfg == 24 || cfg == 25 || cfg == 26 || cfg == 27 || cfg == 28 || cfg == 29)) {
        uia_set_error("uia_gemm: K-blocked activations (a_kb_rows / outT_kb_rows) need a ring tile config with 64-byte sub-tiles (8, 10, 13, 14), not %d", cfg);
        return -1;
    }
    if ((p.resid_lo8 || p.out_lo8) && !(cfg == 8 || cfg == 10 || cfg == 13 || cfg == 14 || cfg == 24 || cfg == 25 || cfg == 26 || cfg == 27 || cfg == 28 || cfg == 29)) {
        uia_set_error("uia_gemm: three-byte tensors (resid_lo8 / out_lo8) are read and written by the LDS-patch epilogue of the ring tile configs (8, 10, 13, 14, 24), not %d", cfg);
        return -1;
    }
    if (p.w_kblocked && !ring) {
        uia_set_error("uia_gemm: a K-blocked W needs a ring tile config (8, 9, 10, 13), not %d", cfg);
        return -1;
    }
    if (ring) {
        // Tile order (measured at M = 50 432 / 65 536, profiles/r02_a_gemm_order_layout.txt): a weight that does not fit the XCD's 4 MiB L2
        // beside the A panels (N = 2304: 3.5 MB, N = 3072: 4.7 MB at K = 768) wants the XCD's ~32 concurrent tiles arranged as a block,
        // 8 row panels deep for N = 2304 (+8 %), 16 for N = 3072 (+3 %); for N = 768 the whole W stays resident and the plain order wins.
        int gm = gm_req == 255 ? 0 : gm_req;
        // Round 3 (rocprofv3 --pmc FETCH_SIZE per group size, M = 65 536, K = 768, profiles/r03_f_tile_group_fetch.txt): fabric reads per launch at
        // N = 3072 are 596 / 578 / 481 / 432 / 648 / 1188 MB for no group / 2 / 4 / 8 / 16 / 32 — and the launch takes 330-334 us with every one
        // of them (as the in-step sweep of round 2 found: 43.7-44.3 ms).  The reads are served on-die and do not pace the K loop; 8 is kept for both
        // widths because it moves the fewest bytes.
        if (gm_req == 0) gm = p.N >= 1536 ? 8 : 0;
        xflags = (xflags & ~255) | gm;
    }
    switch (cfg) {
        case 1: return launch_cfg<T, 256, 256, 2, 4>(stream, p);
        case 2: return launch_cfg<T, 256, 128, 4, 2>(stream, p);
        case 3: return launch_cfg<T, 128, 128, 2, 2>(stream, p);
        case 4: return launch_cfg<T, 256, 64, 4, 1>(stream, p);
        case 5: return launch_cfg<T, 128, 64, 2, 1>(stream, p);
        case 21: return launch_cfg<T, 32, 64, 2, 2>(stream, p);
        case 23:
            if (!wide64_ok(p, (int)sizeof(T))) { uia_set_error("uia_gemm: tile cfg 23 is the bf16 K = 64 stream kernel (alpha, bias, accumulator dropout, one residual of the output's type)"); return -1; }
            return launch_wide64(stream, p);
        case 6: return launch_pp<T, 256, 256, 2, 4>(stream, p);
        case 7: return launch_pp<T, 256, 128, 4, 2>(stream, p);
        case 8: return launch_ring<T, 256, 256, 2, 4, 64, 4>(stream, p, true, xflags);
        case 9: return launch_ring<T, 256, 128, 4, 2, 128, 3>(stream, p, false, xflags);
        case 10: return launch_ring<T, 256, 256, 2, 4, 64, 4>(stream, p, false, xflags);   // cfg 8 with the run-time (generic) epilogue: parity cross-check
        case 12: return launch_persist<T>(stream, p);
        case 16:
            if (!skinny64_ok(p, (int)sizeof(T))) { uia_set_error("uia_gemm: tile cfg 16 is the bf16 N = 64 stream kernel (bias + T output only)"); return -1; }
            return launch_skinny64(stream, p);
#ifdef UIA_GEMM_EXP
        case 15: return launch_ring<T, 256, 256, 2, 4, 64, 4, 1>(stream, p, true, xflags);  // free-running loop (two fragment sets, one barrier per sub-tile):
                                                                                            // 5-10 % SLOWER than the ping-pong loop on every shape; experiment builds only
#endif
#ifdef UIA_GEMM_EXP
        case 19: return launch_ring<T, 256, 256, 2, 4, 64, 4, 3>(stream, p, true, xflags);   // cfg 8 with the last 8 MFMAs of a cluster behind the slot barrier
        case 20: return launch_ring<T, 256, 256, 2, 4, 64, 4, 4>(stream, p, true, xflags);   // ... the last 16
#endif
#if defined(UIA_GEMM_EXP) || defined(UIA_GEMM_CFG1718)
        // experiment (round 3): FOUR-wave workgroups, two per CU, so that one workgroup's epilogue runs beside the other's K loop
        case 17: return launch_ring<T, 256, 128, 2, 2, 64, 3>(stream, p, true, xflags);   // 256 x 128 tiles (A panel re-read by the column neighbour)
        case 18: return launch_ring<T, 128, 256, 1, 4, 64, 3>(stream, p, true, xflags);   // 128 x 256 tiles
#endif
        case 24: return launch_ring<T, 256, 256, 2, 4, 64, 5>(stream, p, true, xflags);   // cfg 8 on a 5-deep ring (160 KB of LDS, four sub-tiles in flight): +4-5 % on long-K shapes in isolation, level inside the step: opt-in (ops.RING5)
        case 25:                                                                          // four waves of 128 x 128 (gemm_quad.hip); 26: its run-time epilogue
        case 26:
            if (sizeof(T) != 2) { uia_set_error("uia_gemm: tile cfg %d is bf16 only", cfg); return -1; }
            return uia_gemm_quad_launch(stream, p, cfg == 25, (xflags & 255) | (((cfg_in >> 16) & 7) << 8));   // bits 16-18: K-loop ablation (tools/time_quad.py --diag)
        case 27:                                                                          // four waves of 128 x 128, operands through registers (gemm_quadv.hip): two sub-tiles in flight; 28: three
        case 28:
        case 29:                                                                          // 29: cfg 27 on a persistent grid, the next tile's first sub-tiles requested under the epilogue
            if (sizeof(T) != 2) { uia_set_error("uia_gemm: tile cfg %d is bf16 only", cfg); return -1; }
            return uia_gemm_quadv_launch(stream, p, true, xflags & 255, cfg == 27 ? 2 : cfg == 28 ? 3 : 9);
        case 14: return launch_ring<T, 128, 256, 2, 4, 64, 3>(stream, p, true, xflags);   // 3-deep ring: 72 KB of LDS, two workgroups per CU
        case 13: return launch_ring<T, 128, 256, 2, 4, 64, 4>(stream, p, true, xflags, sk_info);   // half-height tiles: the M tail of a launch whose last round
                                                                                         // would leave most CUs idle (host splits the rows, ops.gemm)
        default: uia_set_error("uia_gemm: unknown tile config %d", cfg); return -1;
    }
}

}  // namespace

int uia_gemm_launch(hipStream_t stream, int dtype, const UiaGemmParams& p, int cfg) {
    const int esz = dtype == UIA_BF16 ? 2 : 4;
    const int bk = 128 / esz;
    UIA_CHECK_ARG(dtype == UIA_BF16 || dtype == UIA_F32, "uia_gemm: bad dtype %d", dtype);
    UIA_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "uia_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    UIA_CHECK_ARG(p.K % bk == 0, "uia_gemm: K=%d must be a multiple of %d for this dtype", p.K, bk);
    UIA_CHECK_ARG(p.N % 8 == 0, "uia_gemm: N=%d must be a multiple of 8", p.N);
    UIA_CHECK_ARG(p.A && p.W, "uia_gemm: null operand");
    UIA_CHECK_ARG((p.a_kb_rows || p.lda >= p.K - (p.K2 > 0 ? p.K2 : 0)) && p.ldw >= p.K, "uia_gemm: leading dimension smaller than K");
    UIA_CHECK_ARG((p.a_kb_rows || (p.lda * esz) % 16 == 0) && (p.ldw * esz) % 16 == 0, "uia_gemm: rows must be 16-byte aligned");
    UIA_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "uia_gemm: operands must be 16-byte aligned");
    UIA_CHECK_ARG(p.outT || p.out32, "uia_gemm: no output");
    UIA_CHECK_ARG(!p.outT || ((p.outT_kb_rows || p.ldo % 8 == 0) && (uintptr_t)p.outT % 16 == 0), "uia_gemm: outT alignment");
    UIA_CHECK_ARG(!p.out32 || (p.ldo32 % 4 == 0 && (uintptr_t)p.out32 % 16 == 0), "uia_gemm: out32 alignment");
    UIA_CHECK_ARG(!p.resid || (p.ldr % 4 == 0 && (uintptr_t)p.resid % 16 == 0), "uia_gemm: resid alignment");
    UIA_CHECK_ARG(!p.residT || ((p.residT_kb_rows || p.ldrT % 8 == 0) && (uintptr_t)p.residT % 16 == 0), "uia_gemm: residT alignment");
    // three-byte tensors: bf16 launches on the LDS-patch epilogues (ring tile configs); the hi planes are residT / outT
    UIA_CHECK_ARG(!p.resid_lo8 || (dtype == UIA_BF16 && p.residT && !p.resid && (p.resid_lo_kb_rows ? (p.resid_lo_kb_rows >= p.M && p.N % 64 == 0) : (p.ld_resid_lo >= p.N && p.ld_resid_lo % 8 == 0)) && (uintptr_t)p.resid_lo8 % 8 == 0 &&
                                  p.resid_mod == 0 && p.out_group == 0),
                  "uia_gemm: resid_lo8 needs bf16, residT as the hi plane, no fp32 resid, 8-byte aligned rows of at least N bytes and no row remapping");
    UIA_CHECK_ARG(p.residT_kb_rows == 0 || (p.resid_lo8 && p.residT_kb_rows >= p.M && (p.N * esz) % 64 == 0), "uia_gemm: residT_kb_rows needs resid_lo8, at least M rows and N*2 a multiple of 64");
    UIA_CHECK_ARG(!p.out_lo8 || (dtype == UIA_BF16 && p.outT && (p.out_lo_kb_rows ? (p.out_lo_kb_rows >= p.M && p.N % 64 == 0) : (p.ld_out_lo >= p.N && p.ld_out_lo % 8 == 0)) && (uintptr_t)p.out_lo8 % 8 == 0 && p.out_group == 0),
                  "uia_gemm: out_lo8 needs bf16, outT as the hi plane, 8-byte aligned rows of at least N bytes and no row remapping");
    UIA_CHECK_ARG(!p.bias || (uintptr_t)p.bias % 16 == 0, "uia_gemm: bias alignment");
    UIA_CHECK_ARG(!p.dact || p.aux_in, "uia_gemm: dact needs aux_in");
    UIA_CHECK_ARG(!p.aux_in || (p.ldaux_in % 8 == 0 && (uintptr_t)p.aux_in % 16 == 0), "uia_gemm: aux_in alignment");
    UIA_CHECK_ARG(!p.aux_out || (p.ldaux_out % 8 == 0 && (uintptr_t)p.aux_out % 16 == 0), "uia_gemm: aux_out alignment");
    UIA_CHECK_ARG(p.resid_mod == 0 || p.resid, "uia_gemm: resid_mod without resid");
    UIA_CHECK_ARG(!p.resid_ln_stats || ((p.resid || p.resid_lo8) && p.resid_ln_w && p.resid_ln_b), "uia_gemm: resid_ln_stats needs resid (or residT + resid_lo8), resid_ln_w and resid_ln_b");
    UIA_CHECK_ARG(!p.resid_ln_stats || ((uintptr_t)p.resid_ln_stats % 8 == 0 && (uintptr_t)p.resid_ln_w % 16 == 0 && (uintptr_t)p.resid_ln_b % 16 == 0),
                  "uia_gemm: resid_ln alignment");
    UIA_CHECK_ARG(p.a_kb_rows == 0 || p.a_kb_rows >= p.M, "uia_gemm: a_kb_rows=%lld < M=%d", (long long)p.a_kb_rows, p.M);
    UIA_CHECK_ARG(p.outT_kb_rows == 0 || (p.outT && p.outT_kb_rows >= p.M && (p.N * esz) % 64 == 0 && p.out_group == 0),
                  "uia_gemm: outT_kb_rows needs outT, at least M rows, N*sizeof(T) a multiple of 64 and no row remapping");
    UIA_CHECK_ARG(!p.resid_ln_stats || p.resid_ln_dim >= 0, "uia_gemm: resid_ln_dim=%d", p.resid_ln_dim);
    UIA_CHECK_ARG(!p.rowsum_out || ((uintptr_t)p.rowsum_out % 16 == 0 && p.out_group == 0), "uia_gemm: rowsum_out must be 16-byte aligned and takes no row remapping");
    UIA_CHECK_ARG(!(p.resid_ln_stats && p.resid_ln_dim > 0) || (uintptr_t)p.resid_ln_stats % 16 == 0, "uia_gemm: row sums behind resid_ln_stats must be 16-byte aligned");
    UIA_CHECK_ARG(!p.ln_flag || (uintptr_t)p.ln_flag % 4 == 0, "uia_gemm: ln_flag must be a 4-byte aligned device word");
    UIA_CHECK_ARG(p.drop_where >= 0 && p.drop_where <= 2, "uia_gemm: drop_where=%d (0 none, 1 A operand, 2 accumulator)", p.drop_where);
    UIA_CHECK_ARG(p.drop_where == 0 || (p.drop_p >= 0.f && p.drop_p < 1.f), "uia_gemm: drop_p=%f outside [0, 1)", (double)p.drop_p);
    UIA_CHECK_ARG(p.drop_where != 1 || (dtype == UIA_BF16 && (size_t)p.M * (size_t)p.K / 8 <= 0xFFFFFFFFull && (uintptr_t)p.a_drop_out % 16 == 0 && !p.a_kb_rows),
                  "uia_gemm: dropout on the A operand needs bf16, row-major A, M*K <= 2^35 and a 16-byte aligned a_drop_out");
    UIA_CHECK_ARG(p.drop_where != 2 || (size_t)p.M * (size_t)p.N / 8 <= 0xFFFFFFFFull, "uia_gemm: dropout on the accumulator needs M*N <= 2^35");
    UIA_CHECK_ARG(p.drop_where == 1 || !p.a_drop_out, "uia_gemm: a_drop_out without drop_where = 1");
    UIA_CHECK_ARG(p.K2 >= 0 && (p.K2 == 0 || (dtype == UIA_BF16 && p.A2 && p.K2 % 32 == 0 && p.K2 < p.K && p.lda2 >= p.K2 && (p.lda2 * 2) % 16 == 0 && (uintptr_t)p.A2 % 16 == 0 &&
                                             p.a2_group_cols >= 0 && p.a2_group_cols % 256 == 0 && !p.a_kb_rows && p.lda >= p.K - p.K2)),
                  "uia_gemm: the K extension needs bf16, a row-major A of K - K2 columns, a 16-byte aligned row-major A2 with K2 (a multiple of 32, < K) columns and "
                  "column groups that are multiples of 256; got K=%d K2=%d lda2=%lld group=%d", p.K, p.K2, (long long)p.lda2, p.a2_group_cols);
    UIA_CHECK_ARG(!p.lnfold_sums || (p.lnfold_colsum && p.lnfold_dim > 0 && p.alpha == 1.0f && (uintptr_t)p.lnfold_sums % 16 == 0 && (uintptr_t)p.lnfold_colsum % 16 == 0),
                  "uia_gemm: lnfold_sums needs lnfold_colsum (16-byte aligned), lnfold_dim > 0 and alpha == 1");
    // every row the epilogue touches must hold N elements: a leading dimension below N would make row m's tail overwrite row m+1
    UIA_CHECK_ARG(!p.outT || p.outT_kb_rows || p.ldo >= p.N, "uia_gemm: ldo=%lld < N=%d", (long long)p.ldo, p.N);
    UIA_CHECK_ARG(!p.out32 || p.ldo32 >= p.N, "uia_gemm: ldo32=%lld < N=%d", (long long)p.ldo32, p.N);
    UIA_CHECK_ARG(!p.aux_out || p.ldaux_out >= p.N, "uia_gemm: ldaux_out=%lld < N=%d", (long long)p.ldaux_out, p.N);
    UIA_CHECK_ARG(!p.aux_in || p.ldaux_in >= p.N, "uia_gemm: ldaux_in=%lld < N=%d", (long long)p.ldaux_in, p.N);
    UIA_CHECK_ARG(!p.resid || p.ldr >= p.N, "uia_gemm: ldr=%lld < N=%d", (long long)p.ldr, p.N);
    UIA_CHECK_ARG(!p.residT || p.residT_kb_rows || p.ldrT >= p.N, "uia_gemm: ldrT=%lld < N=%d", (long long)p.ldrT, p.N);
    if (dtype == UIA_BF16) return launch_typed<bf16_t>(stream, p, cfg);
    return launch_typed<float>(stream, p, cfg);
}
