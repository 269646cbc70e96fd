// gemm_quad.hip — tile cfg 25: the 256 x 256 output tile on FOUR waves (one per SIMD), each owning 128 x 128 of it.
//
// Why (round 4): the ring kernel (gemm.hip, cfg 8) runs the same tile on eight waves of 128 x 64.  Per 32-deep K step a wave reads
// (128 + 64) rows x 64 bytes of fragments for 32 MFMAs: 196 KB of ds_read per workgroup and step against 2048 cycles of MFMA issue at
// 128 B/clk of LDS — 1536 cycles, plus the 512 of the LDS-DMA writes: the LDS is as busy as the matrix pipe, and the two wave groups meet
// at two barriers per step (in-kernel stamps, round 3: 2738 cycles per 64-deep K step for 2048 of MFMA issue).  With 128 x 128 per wave the
// fragment traffic is (128 + 128) x 64 B for 64 MFMAs: 131 KB per workgroup and step (1024 cycles), one barrier per step, and a wave has
// 64 independent accumulator tiles to keep its SIMD's matrix pipe fed on its own.  This is the shape the vendor library picks for these
// problems (tools/vendor_kernel_names.sh: MT256x256x64, four waves, 256 accumulator registers).
//
// Structure: the ring kernel's operand path unchanged — 64-byte sub-tiles, LDS-DMA (global_load_lds_dwordx4) straight into a 4-deep ring of
// 32 KB buffers, the 4-entry XOR swizzle, row-major or K-blocked operands, the K extension — with a FREE-RUNNING loop: every wave holds two
// fragment sets; in step t it issues the DMA of sub-tile t+4 (into the buffer whose fragments it read one step ago), reads the fragments of
// sub-tile t+1 into the idle set and multiplies the set it read in step t-1: its own ds_reads and DMA issue ride in the shadow of its own
// MFMAs.  256 accumulator registers + 2 x 64 fragment registers: one wave per SIMD (512-register budget).
// The epilogue is gemm_epilogue_lds (gemm_epilogue.h) run twice, once per 64-column half of the wave tile: same arithmetic in the same
// order per element as cfg 8, and the K loop adds the same products in the same order: results are bit-identical to cfg 8's.
#include "gemm_epilogue.h"
#include <type_traits>

namespace {

// DIAG (experiment builds of the plain-store mask, tools/time_quad.py --diag): 1 = K loop without its LDS-DMA after the prologue, 2 = without its
// fragment reads, 4 = without its MFMAs.  Results are garbage; only the time is read.
template <int EPI, bool A2X, int DIAG = 0>
__global__ __launch_bounds__(256, 1) void gemm_tn_quad_kernel(const UiaGemmParams p, const int xflags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = bf16_t;
    constexpr int BM = 256, BN = 256, NW = 4, WTM = 128, WTN = 128, MT = 8, NT = 8, BKB = 64, NBUF = 4, ESZ = 2;
    constexpr int A_BYTES = BM * BKB, W_BYTES = BN * BKB, BUF_BYTES = A_BYTES + W_BYTES;
    constexpr int RPI = 1024 / BKB, CPR = BKB / 16;                     // rows per 1 KiB DMA piece, 16-byte chunks per row
    constexpr int A_PER_WAVE = (BM / RPI) / NW, W_PER_WAVE = (BN / RPI) / NW, GPT = A_PER_WAVE + W_PER_WAVE;
    static_assert(GPT == 8, "eight 1 KiB pieces per wave and sub-tile");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // workgroup -> tile: each XCD (blockIdx & 7) walks one contiguous run of tiles; inside it groups of GM row panels, column by column
    const int tiles_n = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
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

    // row statistics the epilogue wants (folded / deferred LayerNorm): requested ahead of the K loop, as in the ring kernel
    constexpr bool LNROW = EPI == EPI_GENERIC || (EPI & (EPI_LNFOLD | EPI_RESID_LN)) != 0;
    constexpr int LNR = WTM / 64;
    float2 lnpre[LNR];
#pragma unroll
    for (int i = 0; i < LNR; ++i) lnpre[i] = float2{0.f, 1.f};
    if (LNROW && p.lnfold_sums) {
#pragma unroll
        for (int i = 0; i < LNR; ++i) {
            const int m = m0 + wm * WTM + lane + 64 * i;
            if (m < p.M) lnpre[i] = rowsum_load(p.lnfold_sums, (size_t)m);
        }
    } else if (LNROW && p.resid_ln_stats && p.resid_mod == 0 && p.out_group == 0) {
#pragma unroll
        for (int i = 0; i < LNR; ++i) {
            const int m = m0 + wm * WTM + lane + 64 * i;
            if (m < p.M) lnpre[i] = p.resid_ln_dim > 0 ? rowsum_load(p.resid_ln_stats, (size_t)m) : *(const float2*)((const float*)p.resid_ln_stats + 2 * (size_t)m);
        }
    }

    const bool kbA = p.a_kb_rows != 0, kbW = p.w_kblocked != 0;
    auto swzA = [](int r) -> int { return (0x1230 >> (4 * ((r >> 2) & 3))) & 3; };
    auto swzW = [](int rl) -> int { return (0x1230 >> (4 * ((rl >> 4) & 3))) & 3; };
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
        const int c = (lane % CPR) ^ swzW(r & 63);
        int gn = n0 + r;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = kbW ? (const char*)p.W + (size_t)gn * BKB + c * 16 : (const char*)p.W + ((size_t)gn * (size_t)p.ldw) * ESZ + c * 16;
    }
    const size_t kstepA = kbA ? (size_t)p.a_kb_rows * BKB : (size_t)BKB, kstepW = kbW ? (size_t)p.N * BKB : (size_t)BKB;
    const char* srcA2[A2X ? A_PER_WAVE : 1];
    int ntl1 = 0;
    if constexpr (A2X) {                                   // K extension: the last K2 columns of the K loop come from A2 (gemm.hip, ring kernel)
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
    // fragment addresses inside a ring buffer.  A (the MFMA B operand): row wm·128 + 16·mt + li, chunk g.  W (the MFMA A operand): the rows of
    // each 64-column half are read in the permuted order 16·(li>>2) + 4·j + (li&3), so a lane ends up with 16 consecutive columns of a row.
    const int li = lane & 15, g = lane >> 4;
    const int offA0 = (wm * WTM + li) * BKB + ((g ^ swzA(li)) << 4);
    const int offW0 = A_BYTES + (wn * WTN + (li >> 2) * 16 + (li & 3)) * BKB + ((g ^ swzW((li >> 2) * 16)) << 4);

    f32x4 acc[2][MT][4];                                   // [column half][row group][column group]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntl = (p.K * ESZ) / BKB;
    auto stage_piece = [&](int t, int i) {                 // piece i (0..7) of this wave's share of sub-tile t
        char* base = smem + (t & (NBUF - 1)) * BUF_BYTES;
        if (i < A_PER_WAVE) {
            if (A2X && t >= ntl1) glds16_asm(srcA2[i] + (size_t)(t - ntl1) * BKB, base + (wave + NW * i) * 1024);
            else glds16_asm(srcA[i] + (size_t)t * kstepA, base + (wave + NW * i) * 1024);
        } else {
            glds16_asm(srcW[i - A_PER_WAVE] + (size_t)t * kstepW, base + A_BYTES + (wave + NW * (i - A_PER_WAVE)) * 1024);
        }
    };
    // prologue: four sub-tiles in flight; sub-tiles 0 and 1 resident before the first fragment reads
    for (int t = 0; t < NBUF && t < ntl; ++t)
#pragma unroll
        for (int i = 0; i < GPT; ++i) stage_piece(t, i);
    if (ntl >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPT) : "memory");
    else if (ntl == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // The accumulators are pinned to the AGPR half of the register file and the fragments to the VGPR half through the operand constraints of
    // the MFMA itself: left to the allocator (builtin form) the 256 + 128 live registers came out as fragments in AGPRs and accumulators
    // shuttled through v_accvgpr moves with 320 spilled registers.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto mma = [](f32x4& c, const u32x4& w, const u32x4& a) { asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a)); };
    u32x4 af0[MT], wf0[NT], af1[MT], wf1[NT];
    auto read_a = [&](u32x4 (&a_)[MT], const char* buf, int i) { a_[i] = *(const u32x4*)(buf + offA0 + i * 16 * BKB); };
    auto read_w = [&](u32x4 (&w_)[NT], const char* buf, int j) { w_[j] = *(const u32x4*)(buf + offW0 + (j >> 2) * 64 * BKB + (j & 3) * 4 * BKB); };
#pragma unroll
    for (int j = 0; j < NT; ++j) read_w(wf0, smem, j);
#pragma unroll
    for (int i = 0; i < MT; ++i) read_a(af0, smem, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // step 0 refills buffer 0: every wave's reads of it must be back first
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // One step = eight groups of [one DMA piece of sub-tile t+4 | two fragment reads of sub-tile t+1 | the eight MFMAs of one row group]:
    // the order is pinned group by group so that the reads and the DMA issue are spread under the whole MFMA chain.  DMA = false: the last
    // steps of the K loop (nothing left to request; their fragment reads past the last sub-tile land in registers nobody multiplies).
    auto step = [&](auto dma_c, int t, u32x4 (&ac)[MT], u32x4 (&wc)[NT], u32x4 (&an)[MT], u32x4 (&wn_)[NT]) {
        constexpr bool DMA = decltype(dma_c)::value;
        const char* nbuf = smem + ((t + 1) & (NBUF - 1)) * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (DMA && !(DIAG & 1)) stage_piece(t + NBUF, i);
            if constexpr (!(DIAG & 2)) { read_w(wn_, nbuf, i); read_a(an, nbuf, i); }
            if constexpr (!(DIAG & 4)) {
#pragma unroll
                for (int j = 0; j < NT; ++j) mma(acc[j >> 2][i][j & 3], wc[j], ac[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DMA && !(DIAG & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPT) : "memory");   // own pieces of sub-tile t+2 landed; t+3 and t+4 stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    for (; t + 1 + NBUF < ntl; t += 2) {
        step(std::true_type{}, t, af0, wf0, af1, wf1);
        step(std::true_type{}, t + 1, af1, wf1, af0, wf0);
    }
    for (; t < ntl; t += 2) {
        step(std::false_type{}, t, af0, wf0, af1, wf1);
        if (t + 1 < ntl) step(std::false_type{}, t + 1, af1, wf1, af0, wf0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results are read (v_accvgpr_read) by code the hazard pass cannot relate to them

    // epilogue: the two 64-column halves of the wave tile, each exactly one wave tile of the eight-wave kernel (wave column 2·wn + h)
    float* lnrow = LNROW ? (float*)(smem + NW * EpiPatch<MT, 64>::BYTES_PER_WAVE + wave * (WTM * 8)) : nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && (EPI == EPI_GENERIC || (EPI & EPI_ROWSUM) != 0)) __syncthreads();      // the row-sum exchange of the first half reused the patches
        gemm_epilogue_lds<T, MT, 4, WTM, 64, EPI, false, 2>(p, acc[h], smem, wave, lane, m0, n0, wm, 2 * wn + h, lnrow, LNROW ? lnpre : nullptr);
    }
}

template <int EPI, bool A2X = false, int DIAG = 0>
int launch_quad_epi(hipStream_t stream, const UiaGemmParams& p, int xflags) {
    constexpr bool LNROW = EPI == EPI_GENERIC || (EPI & (EPI_LNFOLD | EPI_RESID_LN)) != 0;
    constexpr int EPB = 4 * EpiPatch<8, 64>::BYTES_PER_WAVE + (LNROW ? 4 * 128 * 8 : 0);
    constexpr int RING = 4 * 512 * 64;
    constexpr int LDS = RING > EPB ? RING : EPB;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = gemm_tn_quad_kernel<EPI, A2X, DIAG>;
    static UiaDevOnce attr_once;
    UIA_ENSURE_LDS_ATTR(attr_once, kern, LDS);
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS, stream, p, xflags);
    UIA_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// bf16 only; the caller (uia_gemm_launch) has validated the descriptor.  specialise = false: the run-time epilogue (parity cross-check).
int uia_gemm_quad_launch(hipStream_t stream, const UiaGemmParams& p, bool specialise, int xflags) {
    if (p.K2 > 0) {
        switch (specialise ? epi_mask_of(p) : EPI_GENERIC) {
            case EPI_BIAS | EPI_OUTT: return launch_quad_epi<(EPI_BIAS | EPI_OUTT), true>(stream, p, xflags);
            case EPI_BIAS | EPI_RESID | EPI_OUT32: return launch_quad_epi<(EPI_BIAS | EPI_RESID | EPI_OUT32), true>(stream, p, xflags);
            default: return launch_quad_epi<EPI_GENERIC, true>(stream, p, xflags);
        }
    }
    if (specialise && epi_mask_of(p) == EPI_OUTT && ((xflags >> 8) & 7)) {         // diagnostic K loops (xflags bits 8-10)
        switch ((xflags >> 8) & 7) {
            case 1: return launch_quad_epi<EPI_OUTT, false, 1>(stream, p, xflags);
            case 2: return launch_quad_epi<EPI_OUTT, false, 2>(stream, p, xflags);
            case 3: return launch_quad_epi<EPI_OUTT, false, 3>(stream, p, xflags);
            case 4: return launch_quad_epi<EPI_OUTT, false, 4>(stream, p, xflags);
            case 5: return launch_quad_epi<EPI_OUTT, false, 5>(stream, p, xflags);
            case 6: return launch_quad_epi<EPI_OUTT, false, 6>(stream, p, xflags);
            default: return launch_quad_epi<EPI_OUTT, false, 7>(stream, p, xflags);
        }
    }
    if (specialise) {
        switch (epi_mask_of(p)) {
#define UIA_EPI_CASE(MASK) case (MASK): return launch_quad_epi<(MASK)>(stream, p, xflags)
            UIA_EPI_CASE(EPI_OUTT);
            UIA_EPI_CASE(EPI_BIAS | EPI_OUTT);
#ifndef UIA_QUAD_FEW
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUT32);
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_OUTT);
            UIA_EPI_CASE(EPI_DGELU | EPI_OUTT);
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT);
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_OUTT);
            UIA_EPI_CASE(EPI_QUICK | EPI_DGELU | EPI_OUTT);
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT);
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUT32 | EPI_OUTT | EPI_ROWSUM);
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID_LO | EPI_RESID_LN | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM);
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID | EPI_OUTT | EPI_OUT_LO | EPI_ROWSUM);
            UIA_EPI_CASE(EPI_BIAS | EPI_RESID_LO | EPI_OUT32);
            UIA_EPI_CASE(EPI_BIAS | EPI_OUTT | EPI_LNFOLD);
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD);
            UIA_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD);
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_OUTT | EPI_LNFOLD);
            UIA_EPI_CASE(EPI_QUICK | EPI_BIAS | EPI_GELU | EPI_AUX_OUT | EPI_OUTT | EPI_LNFOLD);
#endif
#undef UIA_EPI_CASE
            default: break;
        }
    }
    return launch_quad_epi<EPI_GENERIC>(stream, p, xflags);
}
