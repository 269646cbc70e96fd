// attention_bwd.hip — dQ, dK, dV of softmax(q kᵀ·scale + mask) v for short sequences, one
// workgroup per (batch, head); probabilities are recomputed from the forward's log-sum-exp.
//
// The reference has no explicit backward (autograd through nn.MultiheadAttention / SDPA:
// /root/reference/src/third_party/openai_clip/model.py:195-197, src/adapters/lora.py:188);
// the equations are the standard ones:
//     P = exp(S·scale − lse),  dP = dO·Vᵀ,  δ = rowsum(dO ⊙ O),  dS = P ⊙ (dP − δ)·scale,
//     dV = Pᵀ·dO,  dK = dSᵀ·Q,  dQ = dS·K.
//
// bf16 path (MFMA), 8 waves: K of the head is staged once in LDS (row-major, XOR-swizzled); V passes through the dSᵀ buffers into
// the registers of the wave that owns the key tile; Q, dO and O stream through a 4-slot ring of 32-query blocks (LDS-DMA issued
// three blocks ahead, retired with counted vmcnt waits), δ = rowsum(dO ⊙ O) is computed one block ahead from the ring.
//   * S and dP are computed with the KEY on the MFMA lane (A = Q / dO rows, B = K / V rows), so a
//     lane holds P[q = 16·qt + 4g + r][key]; two query tiles give the 8-element B fragment of
//     dVᵀ += dOᵀ·P and dKᵀ += Qᵀ·dS with no lane movement.  The transposed A operands (dOᵀ, Qᵀ)
//     are ds_read_b64_tr_b16 reads of the row-major tiles.
//   * each wave owns key tiles {w, w+8, ...} and keeps their K/V fragments and dKᵀ / dVᵀ in registers for the
//     whole sweep over queries: no cross-workgroup (or cross-wave) reduction for dK, dV.
//   * dS crosses LDS once per 32-query block as dSᵀ (bf16, [keys][32 queries], one 8-byte store per query tile)
//     for dQᵀ = Kᵀ·dSᵀ; both of its operands are transpose reads.
//   * interior tiles of an unmasked head take a path with no per-element tests; exp2 is the raw v_exp_f32.
//     (The first version was VALU-bound at one wave per SIMD: ~20 VALU instructions and 8 two-byte LDS stores per
//     element group, 424 us per ViT-B layer at B=256 against an MFMA floor of ~50.)
// fp32 path (parity mode): plain VALU, three sweeps (dQ per query, dV per key, dK per key).
#include "uia_common.h"
#include "uia_kernels.h"

#ifdef ABWD_STAMPS
// diagnostic build (tools/abwd_stamps.sh): cycles per phase of the bf16 backward, summed over the query blocks of ONE workgroup, per wave:
// [0] wait + barrier at the top of a block, [1] dQ store + LDS-DMA issue + delta, [2] S / dP / dS / dV / dK of the owned key tiles,
// [3] dQ product, [4] prologue (to the first barrier + delta), [5] whole kernel
__device__ unsigned long long uia_abwd_stamps[8 * 8];
extern "C" int uia_abwd_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(uia_abwd_stamps), sizeof(uia_abwd_stamps));
}
#define STAMP() __builtin_amdgcn_s_memtime()
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// LDS-DMA through inline asm: hipcc models the builtin as an LDS store and drains it (s_waitcnt vmcnt(0)) in front of the next LDS
// access, which would serialise the ring below.  The asm form is invisible to that pass; the kernel retires its pieces with its own
// counted s_waitcnt vmcnt(N) + barrier before any read.  M0 carries the wave-uniform LDS byte address of the piece (lane i lands at
// +16·i, or +4·i for the dword form); it is saved and restored around the instruction.
__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    const unsigned dst_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst_u) : "memory");
}
__device__ __forceinline__ void glds4(const char* gsrc, char* lds_wave_base) {
    const unsigned dst_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst_u) : "memory");
}
// sum over the 16 lanes of a DPP row, in every lane: four v_add_f32 with a DPP operand (quad swaps, half-row mirror, row mirror);
// __shfl_xor goes through ds_bpermute, a 4-deep chain of LDS round trips at the top of every block (23 us of the kernel).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}
typedef __attribute__((address_space(3))) const char* lptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ uint4 lds_u4(lptr p) { return __builtin_bit_cast(uint4, *(__attribute__((address_space(3))) const u32x4*)p); }
__device__ __forceinline__ f32x4 lds_f4(lptr p) { return *(__attribute__((address_space(3))) const f32x4*)p; }
__device__ __forceinline__ float lds_f1(lptr p) { return *(__attribute__((address_space(3))) const float*)p; }
__device__ __forceinline__ bf16x8 tr_pair(lptr lo_addr, lptr hi_addr) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo_addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi_addr);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 tr_pair(const char* lo_addr, const char* hi_addr) {
    const s16x4 lo = lds_tr16(lo_addr), hi = lds_tr16(hi_addr);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 mma(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// K image: [rows][128 B], 16-B chunk c of row r stored at chunk c ^ ((r>>1)&7).  Q, dO and O travel through a ring of 32-query
// blocks (one 12 KiB slot = [Q | dO | O] pieces of [32][128 B], same swizzle): a block is only ever needed in "its" iteration
// (S / dP / dV / dK) and, one iteration earlier, for δ = rowsum(dO ⊙ O); the K image and dSᵀ carry everything else.  The whole head
// used to be staged before the first MFMA (75 of the kernel's 313 us with nothing beside it at one workgroup per CU,
// tools/abwd_variants.sh); now K + block 0 are, and block u+3 lands while block u is processed.
// dSᵀ tile: [keys][64 B] (32 queries of the current block), 8-B chunk c of row r stored at chunk c ^ ((r>>1)&7).
constexpr int BWD_WAVES = 8;
constexpr int RING = 4, SLOT = 3 * 32 * 128;
__host__ __device__ constexpr int bwd_lds_bytes(int LPK) { return LPK * 128 + RING * SLOT + 2 * LPK * 64 + 3 * LPK * 4; }

// LDS-DMA and plain loads complete in issue order, so "everything but the last n issued" is a counted wait (stores in flight only
// make it wait longer).
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

template <int LT_MAX>
__global__ __launch_bounds__(64 * BWD_WAVES) void attn_bwd_bf16_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KTW = (LT_MAX + BWD_WAVES - 1) / BWD_WAVES;   // key tiles owned by one wave
    const int L = p.L;
    const int LT = (L + 15) >> 4, NP = (LT + 1) >> 1, LPK = NP * 32;
    char* Ks = smem;
    char* ring = Ks + LPK * 128;
    char* dST0 = ring + RING * SLOT;               // 2 × [LPK keys][32 queries] bf16: block u writes buffer u&1 while dQ of block u-1 reads the other
    float* lraw = (float*)(dST0 + 2 * LPK * 64);   // [LPK] lse as stored by the forward
    float* lse2 = lraw + LPK;                      // [LPK] lse in base-2 units
    float* dls = lse2 + LPK;                       // [LPK] δ·scale

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ABWD_STAMPS
    const unsigned long long st_begin = STAMP();
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_pro = 0;
#endif
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const size_t row0 = (size_t)b * L;
    const size_t rs = (size_t)p.ld_qkv * 2, rso = (size_t)p.lddo * 2, rsO = (size_t)p.ldo * 2;
    const char* qb = (const char*)p.q + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* kb = (const char*)p.k + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* vb = (const char*)p.v + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* gb = (const char*)p.dout + (row0 * p.lddo + (size_t)h * 64) * 2;
    const char* ob = (const char*)p.out + (row0 * p.ldo + (size_t)h * 64) * 2;
    const int li = lane & 15, g = lane >> 4;
    int klen = L;                                          // read before any piece is in flight: the compiler answers this load with vmcnt(0)
    if (p.mask_kind == UIA_MASK_KEYPAD && p.keylen) { klen = p.keylen[b]; klen = klen < 1 ? 1 : (klen > L ? L : klen); }
    asm volatile("" : "+v"(klen));

    // ---- issue order = order of need: K, V (registers), lse, blocks 0..2
    const int ninstr = LPK >> 3;
    for (int q = wave; q < ninstr; q += BWD_WAVES) {
        const int r = 8 * q + (lane >> 3);
        const int gr = r < L ? r : L - 1;
        const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        glds16(kb + gr * rs + c, Ks + q * 1024);
    }
    // V is only ever a B operand of the wave that owns the key tile, for the whole sweep over queries: its image lands in the (still
    // unused) dSᵀ buffers, the fragments are read into registers next to K's, and the first dSᵀ store comes two barriers later.
    // (Plain loads into registers would not do: the compiler answers their first use with vmcnt(0), which drains the ring.)
    for (int q = wave; q < ninstr; q += BWD_WAVES) {
        const int r = 8 * q + (lane >> 3);
        const int gr = r < L ? r : L - 1;
        const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        glds16(vb + gr * rs + c, dST0 + q * 1024);
    }
    for (int i = wave; i * 64 < LPK; i += BWD_WAVES) {
        int r = 64 * i + lane;
        r = r < L ? r : L - 1;
        glds4((const char*)(p.lse + ((size_t)b * p.H + h) * L + r), (char*)(lraw + 64 * i));
    }
    // one block = 12 pieces of 1 KiB: waves 0-3 bring a Q and an O piece each, waves 4-7 a dO piece
    const int pb = wave < 4 ? 2 : 1;
    auto issue_block = [&](int u) {
        char* slot = ring + (u & (RING - 1)) * SLOT;
        const int wq = wave & 3;
        const int r = 8 * wq + (lane >> 3);                      // row within the block
        int gr = 32 * u + r;
        gr = gr < L ? gr : L - 1;
        const int c = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        if (wave < 4) {
            glds16(qb + gr * rs + c, slot + wq * 1024);
            // K-blocked O (the forward wrote it for the output projection): 16-byte chunk j of the head's 128-byte row sits in column
            // block 2h + (j >> 2), at (j & 3)·16 bytes of the row's 64-byte piece
            glds16(p.out_kb_rows ? (const char*)p.out + (((size_t)(2 * h + (c >> 6)) * (size_t)p.out_kb_rows + row0 + gr) << 6) + (c & 63) : ob + gr * rsO + c,
                   slot + 8192 + wq * 1024);
        } else {
            glds16(gb + gr * rso + c, slot + 4096 + wq * 1024);
        }
    };
    issue_block(0);
    if (NP > 1) issue_block(1);
    if (NP > 2) issue_block(2);
    // δ·scale and lse of one block: 16 threads per query row, 8 bytes of dO and of O each
    const float sc = p.scale * 1.44269504088896341f;
    auto delta_block = [&](int ub) {
        const char* slot = ring + (ub & (RING - 1)) * SLOT;
        const int rl = tid >> 4, part = tid & 15;
        const int row = 32 * ub + rl;
        const int off = rl * 128 + ((((part >> 1) ^ (rl >> 1)) & 7) << 4) + 8 * (part & 1);
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        const bf16x4_t gv = *(const bf16x4_t*)(slot + 4096 + off), ov = *(const bf16x4_t*)(slot + 8192 + off);
        float d = 0.f;
#ifndef ABWD_NO_DELTA
#pragma unroll
        for (int e = 0; e < 4; ++e) d = fmaf((float)gv[e], (float)ov[e], d);
#endif
        d = row16_sum(d);
        if (part == 0) {
            dls[row] = row < L ? d * p.scale : 0.f;
            lse2[row] = row < L ? lraw[row] * 1.44269504088896341f : 0.f;
        }
    };
    wait_vm((NP > 2 ? 2 : NP - 1) * pb);                   // K, V, lse and block 0 have landed; blocks 1, 2 stay in flight
    __syncthreads();
    delta_block(0);
#ifdef ABWD_STAMPS
    st_pro = STAMP() - st_begin;
#endif

#ifdef ABWD_PROLOGUE_ONLY
    if (p.L > 0) return;                                  // diagnostic: how long does staging alone take?
#endif
    // row-read fragment offset: row (16·tile + li), chunk g (+4 for the second k-half via ^64)
    const int offR = li * 128 + ((g ^ (li >> 1)) << 4);
    // transpose-read (A operand with k = rows of the image): group g, lane (qq,pp) → row 4g+qq (+16 for hi),
    // columns 16dt + 4pp  → byte 32dt + 8pp → chunk 2dt + (pp>>1)
    const int qq = li >> 2, pp = li & 3;
    const int trow = 4 * g + qq;                        // within a 32-row block (hi half: +16)
    const int tsw = (trow >> 1) & 7;                    // (row>>1)&7; +16 adds 8 → same &7
    // natural-order transpose reads for dQ (k-slot (g,e) ↔ key 8g+e): rows 8g+qq (lo) and 8g+4+qq (hi)
    const int nrow = 8 * g + qq;
    const int nsw_lo = (nrow >> 1) & 7, nsw_hi = ((nrow + 4) >> 1) & 7;

    // K and V fragments of the owned key tiles (B operands of S = Q·Kᵀ and dP = dO·Vᵀ), kept for the whole sweep
    uint4 kf[KTW][2], vf[KTW][2];
#pragma unroll
    for (int a = 0; a < KTW; ++a) {
        const int kt = wave + BWD_WAVES * a;
        const int ktc = kt < 2 * NP ? kt : 0;
        kf[a][0] = *(const uint4*)(Ks + ktc * 2048 + offR);
        kf[a][1] = *(const uint4*)(Ks + ktc * 2048 + (offR ^ 64));
        vf[a][0] = *(const uint4*)(dST0 + ktc * 2048 + offR);
        vf[a][1] = *(const uint4*)(dST0 + ktc * 2048 + (offR ^ 64));
    }
    f32x4 dVt[KTW][4], dKt[KTW][4];
#pragma unroll
    for (int a = 0; a < KTW; ++a)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dVt[a][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dKt[a][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // Software pipeline over the 32-query blocks: iteration u computes S/dP/dS (and dV, dK) of block u into dSᵀ buffer u&1 and, in the
    // same barrier interval, dQ of block u-1 from the other buffer: one barrier per block, and the two kinds of work interleave.
    // dQ of a block is STORED one iteration after it was computed, right behind the barrier and in front of that iteration's LDS-DMA:
    // vmcnt counts stores with the loads, so a store issued after block u+2's pieces made the counted wait at the top of the next
    // iteration ("all but the youngest pb") wait for those pieces too — the ring ran one block shallower than it was built.
    f32x4 dq_hold = f32x4{0.f, 0.f, 0.f, 0.f};
    auto store_dq = [&](int ub) {
        const int hq = wave >> 2, dt = wave & 3;
        const int qrow = 32 * ub + 16 * hq + li;
        if (qrow < L) {
            bf16_t* drow = (bf16_t*)p.dq + (row0 + qrow) * p.ld_dqkv + (size_t)h * 64 + 4 * g;
            if (p.dqkv_kb_rows) drow = (bf16_t*)p.dq + ((size_t)(2 * h + (dt >> 1)) * (size_t)p.dqkv_kb_rows + row0 + qrow) * 32 + 4 * g - 16 * (dt & ~1);
            store4(drow + 16 * dt, dq_hold);
        }
    };
#pragma unroll 1
    for (int u = 0; u <= NP; ++u) {
        char* dST = dST0 + (u & 1) * LPK * 64;
        const char* dSTr = dST0 + ((u + 1) & 1) * LPK * 64;
        const char* Qb = ring + (u & (RING - 1)) * SLOT;
        const char* Gb = Qb + 4096;
        // block u+1 must have landed (its δ is computed now), block u+2 may stay in flight.  The barrier also publishes δ of block u
        // and dSᵀ of block u-1, and frees the slot of block u-1 for block u+3.
#ifdef ABWD_STAMPS
        const unsigned long long st0 = STAMP();
#endif
        if (u + 1 < NP) wait_vm(u + 2 < NP ? pb : 0);
        __syncthreads();
#ifdef ABWD_STAMPS
        const unsigned long long st1 = STAMP();
#endif
        if (u == 0) {                                     // the V image is in registers everywhere: the dSᵀ pad rows (keys past the last tile) become zeros
            for (int i = tid; i < (LPK - 16 * LT) * 4; i += 64 * BWD_WAVES) {
                *(uint4*)(dST0 + 16 * LT * 64 + i * 16) = uint4{0u, 0u, 0u, 0u};
                *(uint4*)(dST0 + LPK * 64 + 16 * LT * 64 + i * 16) = uint4{0u, 0u, 0u, 0u};
            }
        }
#ifndef ABWD_NO_DQ
        if (u >= 2) store_dq(u - 2);
#endif
        if (u + 3 < NP) issue_block(u + 3);
        if (u + 1 < NP) delta_block(u + 1);
#ifdef ABWD_STAMPS
        const unsigned long long st2 = STAMP();
#endif
        if (u < NP) {
        // per-lane query rows of this 32-query block: q(hq, r) = 32u + 16hq + 4g + r
        f32x4 ls[2], dl[2];
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            ls[hq] = *(const f32x4*)(lse2 + 32 * u + 16 * hq + 4 * g);
            dl[hq] = *(const f32x4*)(dls + 32 * u + 16 * hq + 4 * g);
        }
        // A fragments of the two query tiles (Q and dO), both k-halves
        uint4 qf[2][2], gf[2][2];
#pragma unroll
        for (int hq = 0; hq < 2; ++hq)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                qf[hq][kk] = *(const uint4*)(Qb + hq * 2048 + (offR ^ (kk << 6)));
                gf[hq][kk] = *(const uint4*)(Gb + hq * 2048 + (offR ^ (kk << 6)));
            }
        // transposed A operands for dV / dK: dOᵀ and Qᵀ of this block, per d-tile
        // L > 256 (three key tiles per wave: ViT-L/14's 257 tokens): read these eight fragments one d-tile at a time next to the MFMAs
        // that use them — held across the key loop they pushed that instantiation into scratch (38 spilled registers, 615 -> 502 us
        // at B = 256, H = 12); with two tiles per wave holding them is the faster form (298 vs 302 us).
        constexpr bool LAZY_T = LT_MAX > 16;
        bf16x8 gT[4], qT[4];
        if constexpr (!LAZY_T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int off = trow * 128 + (((2 * dt + (pp >> 1)) ^ tsw) << 4) + 8 * (pp & 1);
                gT[dt] = tr_pair(Gb + off, Gb + off + 16 * 128);
                qT[dt] = tr_pair(Qb + off, Qb + off + 16 * 128);
            }
        }
        const bool full_q = p.mask_kind == UIA_MASK_NONE && 32 * u + 32 <= L;
#pragma unroll
        for (int a = 0; a < KTW; ++a) {
            const int kt = wave + BWD_WAVES * a;
            if (kt < LT) {
                const int key = 16 * kt + li;
                bf16x8 pf, sf;
                f32x4 s[2], dp[2];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    s[hq] = mma(qf[hq][0], kf[a][0], f32x4{0.f, 0.f, 0.f, 0.f});
                    dp[hq] = mma(gf[hq][0], vf[a][0], f32x4{0.f, 0.f, 0.f, 0.f});
                }
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    s[hq] = mma(qf[hq][1], kf[a][1], s[hq]);
                    dp[hq] = mma(gf[hq][1], vf[a][1], dp[hq]);
                }
#ifdef ABWD_NO_VALU
                if (true) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { pf[e] = (bf16_t)s[e >> 2][e & 3]; sf[e] = (bf16_t)dp[e >> 2][e & 3]; }
                } else
#endif
                if (full_q && 16 * kt + 16 <= L) {       // interior tile, no mask: no per-element tests
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(fmaf(s[hq][r], sc, -ls[hq][r]));
                            const float dsv = pv * fmaf(dp[hq][r], p.scale, -dl[hq][r]);
                            pf[4 * hq + r] = (bf16_t)pv;
                            sf[4 * hq + r] = (bf16_t)dsv;
                        }
                } else if (p.mask_kind != UIA_MASK_CAUSAL && 32 * u + 32 <= L) {
                    // boundary key tile (or a key-padding mask) under full query rows: ONE select per element on the lane's key.
                    // (The general path below costs ~10 VALU per element; the wave that owns the last, partial key tile of a
                    // 197-token head took it in every block and was the one the other seven waited for: 20.5 K cycles against
                    // 13 K for its two tiles, tools/abwd_stamps.sh.)
                    const bool key_ok = key < klen;
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pe = __builtin_amdgcn_exp2f(fmaf(s[hq][r], sc, -ls[hq][r]));
                            const float pv = key_ok ? pe : 0.f;
                            const float dsv = pv * fmaf(dp[hq][r], p.scale, -dl[hq][r]);
                            pf[4 * hq + r] = (bf16_t)pv;
                            sf[4 * hq + r] = (bf16_t)dsv;
                        }
                } else {
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int qrow = 32 * u + 16 * hq + 4 * g + r;
                            const int kmax = p.mask_kind == UIA_MASK_CAUSAL ? (qrow < klen - 1 ? qrow : klen - 1) : klen - 1;
                            const bool ok = qrow < L && key <= kmax;
                            const float pv = ok ? __builtin_amdgcn_exp2f(fmaf(s[hq][r], sc, -ls[hq][r])) : 0.f;
                            const float dsv = pv * fmaf(dp[hq][r], p.scale, -dl[hq][r]);
                            pf[4 * hq + r] = (bf16_t)pv;
                            sf[4 * hq + r] = (bf16_t)dsv;
                        }
                }
                // dSᵀ row `key`, queries 16hq + 4g .. +3 of the block: one 8-byte store per query tile
                {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                    char* rowp = dST + key * 64;
                    const int sw = (key >> 1) & 7;
                    *(bf16x4_t*)(rowp + (((0 + g) ^ sw) << 3)) = bf16x4_t{sf[0], sf[1], sf[2], sf[3]};
                    *(bf16x4_t*)(rowp + (((4 + g) ^ sw) << 3)) = bf16x4_t{sf[4], sf[5], sf[6], sf[7]};
                }
#ifndef ABWD_NO_DVDK
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if constexpr (LAZY_T) {
                        const int off = trow * 128 + (((2 * dt + (pp >> 1)) ^ tsw) << 4) + 8 * (pp & 1);
                        gT[dt] = tr_pair(Gb + off, Gb + off + 16 * 128);
                        qT[dt] = tr_pair(Qb + off, Qb + off + 16 * 128);
                    }
                    dVt[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT[dt], pf, dVt[a][dt], 0, 0, 0);
                    dKt[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[dt], sf, dKt[a][dt], 0, 0, 0);
                }
#endif
            }
        }
        }
#ifdef ABWD_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long st3 = STAMP();
#endif
        // ---- dQᵀ[d][q] = Σ_key Kᵀ[d][key] · dSᵀ[key][q] of the PREVIOUS block; wave w → query tile hq = w>>2, d-tile w&3.
        //      Both operands are transpose reads: Kᵀ from the K tile, dSᵀ columns from the [key][query] tile.
#ifndef ABWD_NO_DQ
        if (u > 0) {
#else
        if (false) {
#endif
            const int hq = wave >> 2, dt = wave & 3;
            f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f}, dq1 = f32x4{0.f, 0.f, 0.f, 0.f};   // two chains: the 7-9 MFMAs no longer wait on each other
            const int ch = 2 * dt + (pp >> 1);
            const int c8 = 4 * hq + pp;
            auto dq_step = [&](int kbk, f32x4 accq) {
                const char* klo = Ks + (32 * kbk + nrow) * 128 + ((ch ^ nsw_lo) << 4) + 8 * (pp & 1);
                const char* khi = Ks + (32 * kbk + nrow + 4) * 128 + ((ch ^ nsw_hi) << 4) + 8 * (pp & 1);
                const char* slo = dSTr + (32 * kbk + nrow) * 64 + ((c8 ^ nsw_lo) << 3);
                const char* shi = dSTr + (32 * kbk + nrow + 4) * 64 + ((c8 ^ nsw_hi) << 3);
                return __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(klo, khi), tr_pair(slo, shi), accq, 0, 0, 0);
            };
#pragma unroll 1
            for (int kbk = 0; kbk < NP; kbk += 2) {
                dq = dq_step(kbk, dq);
                if (kbk + 1 < NP) dq1 = dq_step(kbk + 1, dq1);
            }
            dq_hold = dq + dq1;
        }
#ifdef ABWD_STAMPS
        asm volatile("" :: "v"(dq_hold));
        const unsigned long long st4 = STAMP();
        st_acc[0] += st1 - st0; st_acc[1] += st2 - st1; st_acc[2] += st3 - st2; st_acc[3] += st4 - st3;
#endif
    }
#ifndef ABWD_NO_DQ
    store_dq(NP - 1);
#endif
    // ---- dK, dV: lane owns key 16kt+li, d = 16dt + 4g + r
#pragma unroll
    for (int a = 0; a < KTW; ++a) {
        const int kt = wave + BWD_WAVES * a;
        const int key = 16 * kt + li;
        if (kt < LT && key < L) {
            bf16_t* krow = (bf16_t*)p.dk + (row0 + key) * p.ld_dqkv + (size_t)h * 64 + 4 * g;
            bf16_t* vrow = (bf16_t*)p.dv + (row0 + key) * p.ld_dqkv + (size_t)h * 64 + 4 * g;
            const size_t kbo = ((size_t)(2 * h) * (size_t)p.dqkv_kb_rows + row0 + key) * 32 + 4 * g, kbp = (size_t)p.dqkv_kb_rows * 32;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // K-blocked: column block 2h + (dt >> 1), 16·(dt & 1) + 4g inside it
                store4(p.dqkv_kb_rows ? (bf16_t*)p.dk + kbo + (dt >> 1) * kbp + 16 * (dt & 1) : krow + 16 * dt, dKt[a][dt]);
                store4(p.dqkv_kb_rows ? (bf16_t*)p.dv + kbo + (dt >> 1) * kbp + 16 * (dt & 1) : vrow + 16 * dt, dVt[a][dt]);
            }
        }
    }
#ifdef ABWD_STAMPS
    if (blockIdx.x == 1500 && lane == 0) {
        unsigned long long* o = uia_abwd_stamps + wave * 8;
        o[0] = st_acc[0]; o[1] = st_acc[1]; o[2] = st_acc[2]; o[3] = st_acc[3]; o[4] = st_pro; o[5] = STAMP() - st_begin;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// Round 4: the barrier-free form.  The kernel above crosses dS through LDS once per 32-query block, so its eight waves meet at a barrier
// seven to nine times per head with 13 / 17 key tiles dealt over them (tools/abwd_stamps.sh: 53 % of the wave cycles parked).  Here no
// product of one wave is ever read by another: a head is 2·LT independent UNITS that the waves of the workgroup pull from an LDS counter —
//   KEY(kt): the wave owns a 16-key tile, keeps its K / V fragments and dKᵀ / dVᵀ in registers and sweeps the 32-query blocks:
//            S = Q·Kᵀ, dP − δ = dO·Vᵀ − δ (key on the lane; −δ is the initial accumulator), P, dS in registers, dVᵀ += dOᵀ·P, dKᵀ += Qᵀ·dS
//   QRY(qt): the wave owns a 16-query tile, keeps its Q / dO fragments and dQᵀ in registers and sweeps the 32-key blocks:
//            S' = K·Qᵀ, dP' − δ = V·dOᵀ − δ (QUERY on the lane, so lse and δ are lane constants), dS' in registers, dQᵀ += Kᵀ·dS'
// i.e. seven MFMA products per (tile, block) pair instead of five and the exponentials twice, for no LDS store, no barrier and no
// lock-step in the sweep.  One barrier per head (after staging).  dS is kept without the softmax scale; dK and dQ take it once at the end.
// LDS: per 16-row tile [Q tile | dO tile] (4 KiB) and [K tile | V tile], so that every read of a block is ONE running address register
// plus an immediate (the first build spent more VALU on LDS addresses than on the softmax); statistics per tile [lse·log2e | −δ].
// Tile swizzle: 16-byte chunk c of row r at chunk c ^ (r & 6) — conflict-free for the ds_read_b128 row reads AND the ds_read_b64_tr_b16
// transposed reads (tools/lds_swizzle_search.py; the (r >> 1) & 7 of the kernels above is 2-way on every transposed read: rows r and
// r + 2 of a 4-row group land in the same banks, a quarter of the LDS cycles in SQ_LDS_BANK_CONFLICT).
// V only ever feeds row reads: the VLDS = false forms take its fragments from global memory (L2) and a 197-token head fits TWICE on a
// CU (2 × 81.5 KB): one head's staging and stores run beside the other's sweep.
struct ctrue { static constexpr bool value = true; };
struct cfalse { static constexpr bool value = false; };
template <int V> struct cint { static constexpr int value = V; };

#ifndef ABWD_ORDER17
#define ABWD_ORDER17 1
#endif
constexpr bool ORDER17 = ABWD_ORDER17 != 0;

template <int NW, bool VLDS, bool OLDS>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void attn_bwd_units_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KVS = VLDS ? 4096 : 2048;                    // bytes per 16-key tile of the K (| V) region
    constexpr bool KGLOBAL = false, QGLOBAL = false;           // every operand fragment comes from the head's LDS images
    const int L = p.L;
    const int LT = (L + 15) >> 4, NP = (LT + 1) >> 1;
    // LDS-address-space pointers: a read is then one running 32-bit register + an immediate (through generic pointers the compiler
    // re-derived every address with a v_add per read)
    char* QGg = smem;                                          // [LT][Q tile 2 KiB | dO tile 2 KiB]
    char* KVg = QGg + LT * 4096;                               // [LT][K tile 2 KiB (| V tile 2 KiB)]
    char* STg = KVg + LT * KVS;                                // [LT][lse·log2e of 16 rows | −δ of 16 rows]  (fp32)
    int* queue = (int*)(STg + LT * 128);
    char* OBg = STg + LT * 128 + 64;                           // OLDS: [LT][O tile] + [16·LT] lse behind it, for δ (ahead of the slack)
    float* lseb = (float*)(OBg + LT * 2048);
    const lptr QG = (lptr)smem, KV = QG + LT * 4096, ST = KV + LT * KVS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ABWD_STAMPS
    // diagnostic build: [0] staging (to the barrier), [1] / [2] cycles in KEY / QRY units, [3] / [4] how many of each, [5] whole kernel
    const unsigned long long st_begin = STAMP();
    unsigned long long st_u[2] = {0, 0}, st_n[2] = {0, 0}, st_pro = 0, st_loop = 0, st_iter = 0;
#endif
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const size_t row0 = (size_t)b * L;
    const size_t rs = (size_t)p.ld_qkv * 2, rso = (size_t)p.lddo * 2, rsO = (size_t)p.ldo * 2;
    const char* qb = (const char*)p.q + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* kb = (const char*)p.k + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* vb = (const char*)p.v + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
    const char* gb = (const char*)p.dout + (row0 * p.lddo + (size_t)h * 64) * 2;
    const char* ob = (const char*)p.out + (row0 * p.ldo + (size_t)h * 64) * 2;
    const int li = lane & 15, g = lane >> 4;
    int klen = L;
    if (p.mask_kind == UIA_MASK_KEYPAD && p.keylen) { klen = p.keylen[b]; klen = klen < 1 ? 1 : (klen > L ? L : klen); }
    klen = __builtin_amdgcn_readfirstlane(klen);
    const bool causal = p.mask_kind == UIA_MASK_CAUSAL;

    // ---- staging.  δ = rowsum(dO ⊙ O) and lse come straight from global memory, eight lanes per row, ALL passes requested before the first
    //      use (one pass at a time was one memory round trip per pass: 21.7 K cycles of staging against 8.4 K); then the images by LDS-DMA
    //      (1 KiB pieces of 8 rows, inverse-swizzled source).
    if constexpr (OLDS) {
        // δ from the LANDED dO image and an O image of its own (same swizzle) instead of a second trip of dO through the memory pipe: a CU pulls
        // ≈ 11 B/clk, and the 26 KB of dO that δ used to re-read were a sixth of the head's staging (the persistent form does the same)
        const int npieces = 2 * LT;
        for (int q = wave; q < npieces; q += NW) {
            const int r = 8 * q + (lane >> 3);
            const int gr = r < L ? r : L - 1;
            const int c = ((lane & 7) ^ (r & 6)) * 16;
            const int t = q >> 1, half = (q & 1) * 1024;
            glds16(gb + gr * rso + c, QGg + t * 4096 + 2048 + half);
            glds16(p.out_kb_rows ? (const char*)p.out + (((size_t)(2 * h + (c >> 6)) * (size_t)p.out_kb_rows + row0 + gr) << 6) + (c & 63) : ob + gr * rsO + c,
                   OBg + q * 1024);
        }
        for (int i = wave; i * 64 < 16 * LT; i += NW) {
            int r = 64 * i + lane;
            r = r < L ? r : L - 1;
            glds4((const char*)(p.lse + ((size_t)b * p.H + h) * L + r), (char*)(lseb + 64 * i));
        }
        for (int q = wave; q < npieces; q += NW) {
            const int r = 8 * q + (lane >> 3);
            const int gr = r < L ? r : L - 1;
            const int c = ((lane & 7) ^ (r & 6)) * 16;
            const int t = q >> 1, half = (q & 1) * 1024;
            glds16(kb + gr * rs + c, KVg + t * KVS + half);
            glds16(qb + gr * rs + c, QGg + t * 4096 + half);
            if (VLDS) glds16(vb + gr * rs + c, KVg + t * KVS + 2048 + half);
        }
        if (tid == 0) *queue = 0;
        // dO, O, lse were requested first: wait for them only (this wave issued nk more pieces behind them), compute δ while K, Q, V land
        const int nk = ((npieces - wave + NW - 1) / NW) * (VLDS ? 3 : 2);
        if (nk >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (nk >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int r0 = 0; r0 < 16 * LT; r0 += NW * 8) {
            const int r = r0 + (tid >> 3), part = tid & 7;
            if (r < 16 * LT) {
                const int off = (r & 15) * 128 + ((part ^ (r & 6)) << 4);
                const bf16x8 gv = *(const bf16x8*)(QGg + (r >> 4) * 4096 + 2048 + off);
                const bf16x8 ovv = *(const bf16x8*)(OBg + (r >> 4) * 2048 + off);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf((float)gv[e], (float)ovv[e], d);
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, false));
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, false));
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xF, 0xF, false));
                if (part == 0) {
                    float* st = (float*)(STg + (r >> 4) * 128) + (r & 15);
                    st[0] = r < L ? lseb[r] * 1.44269504088896341f : 0.f;
                    st[16] = r < L ? -d : 0.f;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else
    {
        constexpr int RPP = NW * 8, MAXP = (288 + RPP - 1) / RPP;
        bf16x8 gv[MAXP], ov[MAXP];
        float lv[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int r = i * RPP + (tid >> 3), part = tid & 7;
            if (i * RPP < 16 * LT) {
                const int gr = r < L ? r : L - 1;
                gv[i] = *(const bf16x8*)(gb + gr * rso + part * 16);
                ov[i] = *(const bf16x8*)(p.out_kb_rows ? (const char*)p.out + (((size_t)(2 * h + (part >> 2)) * (size_t)p.out_kb_rows + row0 + gr) << 6) + (part & 3) * 16
                                                       : ob + gr * rsO + part * 16);
                lv[i] = p.lse[((size_t)b * p.H + h) * L + gr];
            }
        }
        const int npieces = 2 * LT;
        for (int q = wave; q < npieces; q += NW) {
            const int r = 8 * q + (lane >> 3);
            const int gr = r < L ? r : L - 1;
            const int c = ((lane & 7) ^ (r & 6)) * 16;           // tile swizzle: 16-byte chunk c of row r sits at chunk c ^ (r & 6)
            const int t = q >> 1, half = (q & 1) * 1024;
            glds16(kb + gr * rs + c, KVg + t * KVS + half);
            glds16(qb + gr * rs + c, QGg + t * 4096 + half);
            glds16(gb + gr * rso + c, QGg + t * 4096 + 2048 + half);
            if (VLDS) glds16(vb + gr * rs + c, KVg + t * KVS + 2048 + half);
        }
        if (tid == 0) *queue = 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int r = i * RPP + (tid >> 3), part = tid & 7;
            if (i * RPP < 16 * LT) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf((float)gv[i][e], (float)ov[i][e], d);
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xF, 0xF, false));   // row_half_mirror
                if (part == 0 && r < 16 * LT) {
                    float* st = (float*)(STg + (r >> 4) * 128) + (r & 15);
                    st[0] = r < L ? lv[i] * 1.44269504088896341f : 0.f;
                    st[16] = r < L ? -d : 0.f;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#ifdef ABWD_STAMPS
    st_pro = STAMP() - st_begin;
#endif

    const float sc = p.scale * 1.44269504088896341f;
    const int offR0 = li * 128 + ((g ^ (li & 6)) << 4);        // row read: row li of a tile, chunk g; the second k-half is chunk g + 4
    const int offR1 = offR0 ^ 64;
    const int qq = li >> 2, pp = li & 3;
    const int trow = 4 * g + qq;                               // transposed read: group g supplies rows 4g + qq of a tile
    const int tsw = trow & 6;
    int toff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) toff[dt] = trow * 128 + (((2 * dt + (pp >> 1)) ^ tsw) << 4) + 8 * (pp & 1);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    for (;;) {
        int unit = 0;
        if (lane == 0) unit = atomicAdd(queue, 1);
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= 2 * LT) break;
#ifdef ABWD_STAMPS
        const unsigned long long st_u0 = STAMP();
#endif
        // Order of the queue: all KEY units (the longer ones), then all QRY units — except at 17 tiles (ViT-L/14's 257 tokens): 34 units on eight waves are
        // 4.25 rounds, and with three KEY, seven QRY, fourteen KEY, ten QRY the greedy deal ends a unit earlier (list-scheduling model with the measured
        // 8.5 : 5.5 K cycles per unit: makespan 31 against 33.5; at 13 and 16 tiles no order beats KEY-first).
        bool is_key = unit < LT;
        int tile = is_key ? unit : unit - LT;
        if (ORDER17 && LT == 17) {
            if (unit < 3) { is_key = true; tile = unit; }
            else if (unit < 10) { is_key = false; tile = unit - 3; }
            else if (unit < 24) { is_key = true; tile = unit - 7; }
            else { is_key = false; tile = unit - 17; }
        }
        if (is_key) {
            const int kt = tile;
#include "attention_bwd_key_unit.inc"
        } else {
            const int qt = tile;
#include "attention_bwd_qry_unit.inc"
        }
#ifdef ABWD_STAMPS
        st_u[is_key ? 0 : 1] += STAMP() - st_u0; st_n[is_key ? 0 : 1] += 1;
#endif
    }
#ifdef ABWD_STAMPS
    if (blockIdx.x == 1500 && lane == 0 && wave < 8) {
        unsigned long long* o = uia_abwd_stamps + wave * 8;
        o[0] = st_pro; o[1] = st_u[0]; o[2] = st_u[1]; o[3] = st_n[0]; o[4] = st_n[1]; o[5] = STAMP() - st_begin; o[6] = st_loop; o[7] = st_iter;
    }
#endif
}
// ------------------------------------------------------------------------------------------
// The persistent form of the unit kernel: one 8-wave workgroup per CU walks the (batch, head) pairs, and the staging of a head — a third of
// the unit kernel's time, at the ≈ 11 B/clk a CU can pull from memory — runs under the sweeps of its neighbours.  A head is two phases with
// one LDS buffer each:
//   P1(n): KEY units of head n on the Q / dO image of buffer A; meanwhile the K / V image of head n lands in buffer B
//          (the units' own 16-key K / V fragments come straight from global memory: 4 loads per unit);
//   P2(n): QRY units of head n on buffer B; meanwhile the Q / dO image of head n + 1 lands in buffer A and its O rows and lse in regions
//          of their own (the units' Q / dO fragments from global memory); then δ(n + 1) from the landed dO and O images, and P1(n + 1).
// Three barriers per head (end of P1, end of P2, after δ), no spinning on LDS words; statistics are double-buffered.
// (A single queue per head with "buffer B landed" / "buffer A free" words instead of the barrier between the phases was built and measured:
//  the three waves that find no KEY unit in the second round do start QRY units early, and every unit gets 15-40 % slower beside the other
//  kind and the DMA — 227-235 us per ViT-B layer against 265 with the barrier and 238 for the plain unit kernel, which stays the default;
//  tools/abwu_variants.sh: no single ingredient of the sweeps is worth more than 10 % of the kernel.)
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_persist_kernel(const UiaAttnParams p, const int nheads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool VLDS = true, KGLOBAL = true, QGLOBAL = true;
    constexpr int KVS = 4096;
    const int L = p.L;
    const int LT = (L + 15) >> 4, NP = (LT + 1) >> 1;
    char* QGg = smem;                                          // buffer A: [LT][Q tile | dO tile]
    char* KVg = QGg + LT * 4096;                               // buffer B: [LT][K tile | V tile]
    char* STg = KVg + LT * KVS;                                // 2 x [LT][lse·log2e | −δ]
    char* OBg = STg + 2 * LT * 128;                            // [LT][O tile] (same swizzle), for δ of the head that is being staged
    float* lseb = (float*)(OBg + LT * 2048);                   // [16·LT] lse of that head
    int* ctr = (int*)(lseb + ((16 * LT + 63) & ~63));          // unit queues of the two phases (lse lands in whole 64-row pieces)
    const lptr QG = (lptr)smem, KV = QG + LT * 4096;
    const lptr STb = KV + LT * KVS;
    lptr ST = STb;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ABWD_STAMPS
    const unsigned long long st_begin = STAMP();
    unsigned long long st_u[2] = {0, 0}, st_n[2] = {0, 0}, st_pro = 0, st_loop = 0, st_iter = 0, st_bar = 0;
#endif
    const int li = lane & 15, g = lane >> 4;
    const bool causal = p.mask_kind == UIA_MASK_CAUSAL;
    const size_t rs = (size_t)p.ld_qkv * 2, rso = (size_t)p.lddo * 2, rsO = (size_t)p.ldo * 2;
    const float sc = p.scale * 1.44269504088896341f;
    const int offR0 = li * 128 + ((g ^ (li & 6)) << 4);
    const int offR1 = offR0 ^ 64;
    const int qq = li >> 2, pp = li & 3;
    const int trow = 4 * g + qq;
    const int tsw = trow & 6;
    int toff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) toff[dt] = trow * 128 + (((2 * dt + (pp >> 1)) ^ tsw) << 4) + 8 * (pp & 1);
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // the head the units work on ...
    int b = 0, h = 0, klen = L;
    size_t row0 = 0;
    const char *qb = nullptr, *kb = nullptr, *vb = nullptr, *gb = nullptr;
    auto set_head = [&](int n) {
        b = n / p.H; h = n - b * p.H;
        row0 = (size_t)b * L;
        qb = (const char*)p.q + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
        kb = (const char*)p.k + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
        vb = (const char*)p.v + (row0 * p.ld_qkv + (size_t)h * 64) * 2;
        gb = (const char*)p.dout + (row0 * p.lddo + (size_t)h * 64) * 2;
        klen = L;
        if (p.mask_kind == UIA_MASK_KEYPAD && p.keylen) { klen = p.keylen[b]; klen = klen < 1 ? 1 : (klen > L ? L : klen); }
        klen = __builtin_amdgcn_readfirstlane(klen);
    };
    // ... and the staging of a head n (its own address arithmetic: it runs one phase ahead of the units): Q / dO image into buffer A, the O rows
    // and lse that δ needs into their own LDS regions (held in registers across P2 they cost the sweep 25 VGPRs and spills)
    auto stage_qg = [&](int n) {
        const int nb = n / p.H, nh = n - nb * p.H;
        const size_t nrow0 = (size_t)nb * L;
        const char* nq = (const char*)p.q + (nrow0 * p.ld_qkv + (size_t)nh * 64) * 2;
        const char* ng = (const char*)p.dout + (nrow0 * p.lddo + (size_t)nh * 64) * 2;
        const char* no = (const char*)p.out + (nrow0 * p.ldo + (size_t)nh * 64) * 2;
        for (int q = wave; q < 2 * LT; q += NW) {
            const int r = 8 * q + (lane >> 3);
            const int gr = r < L ? r : L - 1;
            const int c = ((lane & 7) ^ (r & 6)) * 16;
            const int t = q >> 1, half = (q & 1) * 1024;
            glds16(nq + gr * rs + c, QGg + t * 4096 + half);
            glds16(ng + gr * rso + c, QGg + t * 4096 + 2048 + half);
            glds16(p.out_kb_rows ? (const char*)p.out + (((size_t)(2 * nh + (c >> 6)) * (size_t)p.out_kb_rows + nrow0 + gr) << 6) + (c & 63) : no + gr * rsO + c,
                   OBg + q * 1024);
        }
        for (int i = wave; i * 64 < 16 * LT; i += NW) {
            int r = 64 * i + lane;
            r = r < L ? r : L - 1;
            glds4((const char*)(p.lse + ((size_t)nb * p.H + nh) * L + r), (char*)(lseb + 64 * i));
        }
    };
    auto stage_kv = [&]() {                                   // K / V image of the CURRENT head into buffer B
        for (int q = wave; q < 2 * LT; q += NW) {
            const int r = 8 * q + (lane >> 3);
            const int gr = r < L ? r : L - 1;
            const int c = ((lane & 7) ^ (r & 6)) * 16;
            const int t = q >> 1, half = (q & 1) * 1024;
            glds16(kb + gr * rs + c, KVg + t * KVS + half);
            glds16(vb + gr * rs + c, KVg + t * KVS + 2048 + half);
        }
    };
    auto delta = [&](char* st_dst) {                          // δ = rowsum(dO ⊙ O) from the landed images: eight lanes per row, 16 bytes of each
        for (int r0 = 0; r0 < 16 * LT; r0 += NW * 8) {
            const int r = r0 + (tid >> 3), part = tid & 7;
            if (r < 16 * LT) {
                const int off = (r & 15) * 128 + ((part ^ (r & 6)) << 4);
                const bf16x8 gv = *(const bf16x8*)(QGg + (r >> 4) * 4096 + 2048 + off);
                const bf16x8 ovv = *(const bf16x8*)(OBg + (r >> 4) * 2048 + off);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf((float)gv[e], (float)ovv[e], d);
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, false));
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, false));
                d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xF, 0xF, false));
                if (part == 0) {
                    float* st = (float*)(st_dst + (r >> 4) * 128) + (r & 15);
                    st[0] = r < L ? lseb[r] * 1.44269504088896341f : 0.f;
                    st[16] = r < L ? -d : 0.f;
                }
            }
        }
    };
    auto next_unit = [&](int which) {
        int u = 0;
        if (lane == 0) u = atomicAdd(ctr + which, 1);
        return __builtin_amdgcn_readfirstlane(u);
    };
    auto seam = [&]() {                                       // everything this wave has in flight (LDS-DMA pieces, stores) has landed; then the workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    int n = blockIdx.x;
    if (n >= nheads) return;
    stage_qg(n);
    if (tid == 0) { ctr[0] = 0; ctr[1] = 0; }
    seam();
    delta(STg);
    __syncthreads();
    set_head(n);
#ifdef ABWD_STAMPS
    st_pro = STAMP() - st_begin;
#endif
    int cur = 0;
    for (;;) {
        ST = STb + cur * (LT * 128);
        // ---- P1: KEY units on buffer A; the head's K / V image lands in buffer B
        stage_kv();
        for (;;) {
            const int kt = next_unit(0);
            if (kt >= LT) break;
#ifdef ABWD_STAMPS
            const unsigned long long st_u0 = STAMP();
#endif
#include "attention_bwd_key_unit.inc"
#ifdef ABWD_STAMPS
            st_u[0] += STAMP() - st_u0; st_n[0] += 1;
#endif
        }
#ifdef ABWD_STAMPS
        const unsigned long long st_b0 = STAMP();
#endif
        seam();                                               // buffer B landed, buffer A free
#ifdef ABWD_STAMPS
        st_bar += STAMP() - st_b0;
#endif
        // ---- P2: QRY units on buffer B; the next head's Q / dO / O images land in buffer A and the O region
        const int nn = n + (int)gridDim.x;
        const bool has_next = nn < nheads;
        if (has_next) stage_qg(nn);
        for (;;) {
            const int qt = next_unit(1);
            if (qt >= LT) break;
#ifdef ABWD_STAMPS
            const unsigned long long st_u0 = STAMP();
#endif
#include "attention_bwd_qry_unit.inc"
#ifdef ABWD_STAMPS
            st_u[1] += STAMP() - st_u0; st_n[1] += 1;
#endif
        }
        if (!has_next) break;
#ifdef ABWD_STAMPS
        const unsigned long long st_b1 = STAMP();
#endif
        seam();                                               // buffer A landed; buffer B and the statistics of head n are free
        delta(STg + (cur ^ 1) * (LT * 128));
        if (tid == 0) { ctr[0] = 0; ctr[1] = 0; }
        __syncthreads();
#ifdef ABWD_STAMPS
        st_bar += STAMP() - st_b1;
#endif
        n = nn;
        set_head(n);
        cur ^= 1;
    }
#ifdef ABWD_STAMPS
    if (blockIdx.x == 100 && lane == 0 && wave < 8) {
        unsigned long long* o = uia_abwd_stamps + wave * 8;
        o[0] = st_pro; o[1] = st_u[0]; o[2] = st_u[1]; o[3] = st_n[0]; o[4] = st_n[1]; o[5] = STAMP() - st_begin; o[6] = st_bar; o[7] = st_iter;
    }
#endif
}
__host__ __device__ constexpr int persist_lds_bytes(int LT) {      // the regions behind buffer B absorb the pipeline's reads one tile past its end
    return LT * 8192 + 2 * LT * 128 + LT * 2048 + ((16 * LT + 63) / 64) * 256 + 64 + 1024;
}
constexpr int PERSIST_LT_MAX = 15;                                  // 240 tokens: 159.7 KB

// + slack behind the statistics: the pipeline requests one block ahead without asking whether its second tile exists, so the bytes of one
// (non-existent) K / V tile behind the K region and of a few statistics rows must lie inside the allocation (they are never used)
__host__ __device__ constexpr int units_lds_bytes(int LT, bool vlds, bool olds = false) {
    const int kvs = vlds ? 4096 : 2048, need = kvs + 64 - LT * 128;
    return LT * 4096 + LT * kvs + LT * 128 + 64 + (olds ? LT * 2048 + ((16 * LT + 63) / 64) * 256 + 1024 : (need > 1024 ? need : 1024));
}
constexpr int OLDS_LT_MAX = 15;                     // O image in LDS up to 240 tokens (13 tiles: 138 KB)


// ------------------------------------------------------------------------------------------
// fp32 parity path.  Sweep 0: dQ (thread per query; K,V in LDS).  Sweeps 1,2: dV then dK
// (thread per key; Q,dO in LDS).
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = p.L;
    float* A = (float*)smem;                 // [L][64]
    float* Bm = A + (size_t)L * 64;          // [L][64]
    float* lse = Bm + (size_t)L * 64;        // [L]
    float* delta = lse + L;                  // [L]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
    const size_t row0 = (size_t)b * L;
    const float* qb = (const float*)p.q + row0 * p.ld_qkv + (size_t)h * 64;
    const float* kb = (const float*)p.k + row0 * p.ld_qkv + (size_t)h * 64;
    const float* vb = (const float*)p.v + row0 * p.ld_qkv + (size_t)h * 64;
    const float* gb = (const float*)p.dout + row0 * p.lddo + (size_t)h * 64;
    const float* ob = (const float*)p.out + row0 * p.ldo + (size_t)h * 64;
    int klen = L;
    if (p.mask_kind == UIA_MASK_KEYPAD && p.keylen) { klen = p.keylen[b]; klen = klen < 1 ? 1 : (klen > L ? L : klen); }

    for (int r = tid; r < L; r += 256) {
        float d = 0.f;
        for (int c = 0; c < 64; ++c) d = fmaf(gb[(size_t)r * p.lddo + c], ob[(size_t)r * p.ldo + c], d);
        delta[r] = d;
        lse[r] = p.lse[((size_t)b * p.H + h) * L + r];
    }
    for (int i = tid; i < L * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        *(f32x4*)(A + r * 64 + c) = *(const f32x4*)(kb + (size_t)r * p.ld_qkv + c);
        *(f32x4*)(Bm + r * 64 + c) = *(const f32x4*)(vb + (size_t)r * p.ld_qkv + c);
    }
    __syncthreads();
    // ---- sweep 0: dQ[q] = Σ_k ds·K[k]
    for (int qi = tid; qi < L; qi += 256) {
        float q[64], go[64], dq[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) { q[c] = qb[(size_t)qi * p.ld_qkv + c]; go[c] = gb[(size_t)qi * p.lddo + c]; dq[c] = 0.f; }
        const int kend = p.mask_kind == UIA_MASK_CAUSAL ? (qi + 1 < klen ? qi + 1 : klen) : klen;
        const float lq = lse[qi], dq_delta = delta[qi];
        for (int k = 0; k < kend; ++k) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) { s = fmaf(q[c], A[k * 64 + c], s); dp = fmaf(go[c], Bm[k * 64 + c], dp); }
            const float pv = expf(s * p.scale - lq);
            const float ds = pv * (dp - dq_delta) * p.scale;
#pragma unroll
            for (int c = 0; c < 64; ++c) dq[c] = fmaf(ds, A[k * 64 + c], dq[c]);
        }
        float* drow = (float*)p.dq + (row0 + qi) * p.ld_dqkv + (size_t)h * 64;
#pragma unroll
        for (int c = 0; c < 64; ++c) drow[c] = dq[c];
    }
    __syncthreads();
    for (int i = tid; i < L * 16; i += 256) {
        const int r = i >> 4, c = (i & 15) * 4;
        *(f32x4*)(A + r * 64 + c) = *(const f32x4*)(qb + (size_t)r * p.ld_qkv + c);
        *(f32x4*)(Bm + r * 64 + c) = *(const f32x4*)(gb + (size_t)r * p.lddo + c);
    }
    __syncthreads();
    // ---- sweeps 1,2: per key
    for (int ki = tid; ki < L; ki += 256) {
        float kv[64], acc[64];
        float* vrow = (float*)p.dv + (row0 + ki) * p.ld_dqkv + (size_t)h * 64;
        float* krow = (float*)p.dk + (row0 + ki) * p.ld_dqkv + (size_t)h * 64;
        if (ki >= klen) {
#pragma unroll
            for (int c = 0; c < 64; ++c) { vrow[c] = 0.f; krow[c] = 0.f; }
            continue;
        }
        const int qstart = p.mask_kind == UIA_MASK_CAUSAL ? ki : 0;
#pragma unroll
        for (int c = 0; c < 64; ++c) { kv[c] = kb[(size_t)ki * p.ld_qkv + c]; acc[c] = 0.f; }
        for (int qi = qstart; qi < L; ++qi) {       // dV[k] = Σ_q p·dO[q]
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) s = fmaf(A[qi * 64 + c], kv[c], s);
            const float pv = expf(s * p.scale - lse[qi]);
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] = fmaf(pv, Bm[qi * 64 + c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 64; ++c) { vrow[c] = acc[c]; acc[c] = 0.f; }
        float vv[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) vv[c] = vb[(size_t)ki * p.ld_qkv + c];
        for (int qi = qstart; qi < L; ++qi) {       // dK[k] = Σ_q ds·Q[q]
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < 64; ++c) { s = fmaf(A[qi * 64 + c], kv[c], s); dp = fmaf(Bm[qi * 64 + c], vv[c], dp); }
            const float pv = expf(s * p.scale - lse[qi]);
            const float ds = pv * (dp - delta[qi]) * p.scale;
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] = fmaf(ds, A[qi * 64 + c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 64; ++c) krow[c] = acc[c];
    }
}

template <int LT_MAX>
int launch_bf16(hipStream_t stream, const UiaAttnParams& p) {
    const int LT = (p.L + 15) / 16, NP = (LT + 1) / 2, LPK = NP * 32;
    const int lds = bwd_lds_bytes(LPK);
    auto kern = attn_bwd_bf16_kernel<LT_MAX>;
    static UiaDevOnce attr_once;
        constexpr int LPKM = ((LT_MAX + 1) / 2) * 32;
        static_assert(bwd_lds_bytes(LPKM) <= 160 * 1024, "LDS budget");
    UIA_ENSURE_LDS_ATTR(attr_once, kern, bwd_lds_bytes(LPKM));
    hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(64 * BWD_WAVES), lds, stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

template <int NW, bool VLDS, bool OLDS>
int launch_units(hipStream_t stream, const UiaAttnParams& p) {
    const int LT = (p.L + 15) / 16;
    auto kern = attn_bwd_units_kernel<NW, VLDS, OLDS>;
    static UiaDevOnce attr_once;
    constexpr int LTM = OLDS ? OLDS_LT_MAX : 18;
    static_assert(units_lds_bytes(LTM, true, OLDS) <= 160 * 1024, "LDS budget");
    UIA_ENSURE_LDS_ATTR(attr_once, kern, units_lds_bytes(LTM, VLDS, OLDS));
    hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(64 * NW), units_lds_bytes(LT, VLDS, OLDS), stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

int launch_persist(hipStream_t stream, const UiaAttnParams& p) {
    const int LT = (p.L + 15) / 16;
    auto kern = attn_bwd_persist_kernel<8>;
    static UiaDevOnce attr_once;
    static_assert(persist_lds_bytes(PERSIST_LT_MAX) <= 160 * 1024, "LDS budget");
    UIA_ENSURE_LDS_ATTR(attr_once, kern, persist_lds_bytes(PERSIST_LT_MAX));
    int dev = 0, ncu = 0;
    UIA_CHECK_HIP(hipGetDevice(&dev));
    UIA_CHECK_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int nheads = p.B * p.H;
    // one workgroup per CU, and every workgroup the same number of heads where that divides (3072 heads on 256 CUs: 12 each)
    int grid = nheads < ncu ? nheads : ncu;
    const int per = (nheads + grid - 1) / grid;
    grid = (nheads + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * 8), persist_lds_bytes(LT), stream, p, nheads);
    UIA_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int uia_attn_bwd_launch(hipStream_t stream, int dtype, const UiaAttnParams& p, int cfg) {
    UIA_CHECK_ARG(cfg >= 0 && cfg <= 7, "uia_attn_bwd: unknown kernel configuration %d", cfg);
    UIA_CHECK_ARG(dtype == UIA_BF16 || dtype == UIA_F32, "uia_attn_bwd: bad dtype %d", dtype);
    UIA_CHECK_ARG(p.B > 0 && p.H > 0 && p.L > 0, "uia_attn_bwd: empty problem");
    UIA_CHECK_ARG(p.scale > 0.f && p.scale < 3.0e38f, "uia_attn_bwd: scale must be positive and finite (matches the forward's lse), got %g", (double)p.scale);
    UIA_CHECK_ARG(!p.cu_seqlens, "uia_attn_bwd: packed sequences (cu_seqlens) are a forward-only layout");
    UIA_CHECK_ARG((p.out_kb_rows == 0 && p.dqkv_kb_rows == 0) ||
                  (p.dh == 64 && dtype == UIA_BF16 && (p.out_kb_rows == 0 || p.out_kb_rows >= (int64_t)p.B * p.L) && (p.dqkv_kb_rows == 0 || p.dqkv_kb_rows >= (int64_t)p.B * p.L)),
                  "uia_attn_bwd: K-blocked out / dq, dk, dv need the bf16 head-dim-64 path and at least B*L rows");
    if (p.dh != 64) return uia_attn_small_launch(stream, dtype, p, true);   // CLIPSeg decoder heads (d_h = 16)
    UIA_CHECK_ARG(p.q && p.k && p.v && p.out && p.dout && p.lse && p.dq && p.dk && p.dv, "uia_attn_bwd: null tensor");
    const int esz = dtype == UIA_BF16 ? 2 : 4;
    UIA_CHECK_ARG((p.ld_qkv * esz) % 16 == 0 && (p.ldo * esz) % 16 == 0 && (p.lddo * esz) % 16 == 0 && (p.ld_dqkv * esz) % 8 == 0,
                  "uia_attn_bwd: leading dimensions must keep 16-byte rows");
    UIA_CHECK_ARG(((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.out | (uintptr_t)p.dout) % 16 == 0, "uia_attn_bwd: alignment");
    UIA_CHECK_ARG(((uintptr_t)p.dq | (uintptr_t)p.dk | (uintptr_t)p.dv) % 8 == 0, "uia_attn_bwd: output alignment");
    UIA_CHECK_ARG(p.mask_kind != UIA_MASK_KEYPAD || p.keylen, "uia_attn_bwd: key-padding mask needs keylen");
    if (dtype == UIA_F32) {
        UIA_CHECK_ARG(p.L <= 272, "uia_attn_bwd: L=%d exceeds 272", p.L);
        const int lds = (2 * p.L * 64 + 2 * p.L) * 4;
        static UiaDevOnce attr_once;
        UIA_ENSURE_LDS_ATTR(attr_once, attn_bwd_f32_kernel, (2 * 272 * 64 + 2 * 272) * 4);
        hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(p.B * p.H), dim3(256), lds, stream, p);
        UIA_CHECK_LAUNCH();
        return 0;
    }
    UIA_CHECK_ARG(p.L <= 288, "uia_attn_bwd: bf16 path keeps Q, K, dO of a head in LDS: L=%d exceeds 288", p.L);
    const int LT = (p.L + 15) / 16;
    // cfg 0 = the default choice; 1 = the lock-step 8-wave kernel of rounds 1-3; 2 / 3 / 4 = the barrier-free unit kernel with
    // 8 waves and V in LDS / 4 waves and V fragments from global memory (two heads per CU up to 208 tokens) / 8 waves, V from global
    // default: the unit kernel.  Its persistent form (cfg 5) hides the staging and gives the time back in barriers and slower units
    // (265 vs 238 us per ViT-B layer with a barrier between the phases, 227-235 with LDS words instead: DESIGN.md §4 round 4)
    if (cfg == 0) cfg = 2;
    if (cfg == 2) return LT <= OLDS_LT_MAX ? launch_units<8, true, true>(stream, p) : launch_units<8, true, false>(stream, p);
    if (cfg == 6) return launch_units<8, true, false>(stream, p);       // cfg 2 with δ's operands from global memory at every length
    if (cfg == 7) return LT <= OLDS_LT_MAX ? launch_units<12, true, true>(stream, p) : launch_units<12, true, false>(stream, p);   // twelve waves, three per SIMD (<= 168 VGPRs, 25 spilled): 248 / 268 / 353 us against cfg 2's 220 / 217 / 348 (round 4: more waves do not buy what the unit loses; sixteen spill 230 registers)
    if (cfg == 3) return launch_units<4, false, false>(stream, p);
    if (cfg == 4) return launch_units<8, false, false>(stream, p);
    if (cfg == 5) {
        UIA_CHECK_ARG(LT <= PERSIST_LT_MAX, "uia_attn_bwd: the persistent kernel holds at most %d tokens (L = %d)", 16 * PERSIST_LT_MAX, p.L);
        return launch_persist(stream, p);
    }
    if (LT <= 8) return launch_bf16<8>(stream, p);
    if (LT <= 16) return launch_bf16<16>(stream, p);
    return launch_bf16<18>(stream, p);
}
