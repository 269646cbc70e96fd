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

template <int NW, bool VLDS>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_units_kernel(const UiaAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KVS = VLDS ? 4096 : 2048;                    // bytes per 16-key tile of the K (| V) region
    const int L = p.L;
    const int LT = (L + 15) >> 4, NP = (LT + 1) >> 1;
    // LDS-address-space pointers: a read is then one running 32-bit register + an immediate (through generic pointers the compiler
    // re-derived every address with a v_add per read)
    char* QGg = smem;                                          // [LT][Q tile 2 KiB | dO tile 2 KiB]
    char* KVg = QGg + LT * 4096;                               // [LT][K tile 2 KiB (| V tile 2 KiB)]
    char* STg = KVg + LT * KVS;                                // [LT][lse·log2e of 16 rows | −δ of 16 rows]  (fp32)
    int* queue = (int*)(STg + LT * 128);
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
    const uint4 zero_u4 = uint4{0u, 0u, 0u, 0u};

    for (;;) {
        int unit = 0;
        if (lane == 0) unit = atomicAdd(queue, 1);
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= 2 * LT) break;
#ifdef ABWD_STAMPS
        const unsigned long long st_u0 = STAMP();
#endif
        if (unit < LT) {
            // =============================== KEY unit: key tile kt, sweep over 32-query blocks
            const int kt = unit;
            const int key = 16 * kt + li;
            uint4 kf[2], vf[2];
            kf[0] = lds_u4(KV + kt * KVS + offR0);
            kf[1] = lds_u4(KV + kt * KVS + offR1);
            if (VLDS) {
                vf[0] = lds_u4(KV + kt * KVS + 2048 + offR0);
                vf[1] = lds_u4(KV + kt * KVS + 2048 + offR1);
            } else {
                const int gk = key < L ? key : L - 1;
                vf[0] = *(const uint4*)(vb + gk * rs + g * 16);
                vf[1] = *(const uint4*)(vb + gk * rs + g * 16 + 64);
            }
            f32x4 dVt[4], dKt[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dVt[dt] = zero4; dKt[dt] = zero4; }
            const int ub_lo = 16 * kt >= klen ? NP : (causal ? kt >> 1 : 0);      // a fully padded key tile: zeros
            const bool key_ok = key < klen;
            // running addresses of the current block: rows (two k-halves), transposed reads (four d-tiles), statistics
            lptr r0 = QG + ub_lo * 8192 + offR0;
            lptr r1 = QG + ub_lo * 8192 + offR1;
            lptr tp0 = QG + ub_lo * 8192 + toff[0];
            lptr tp1 = QG + ub_lo * 8192 + toff[1];
            lptr tp2 = QG + ub_lo * 8192 + toff[2];
            lptr tp3 = QG + ub_lo * 8192 + toff[3];
            lptr sp = ST + ub_lo * 256 + 16 * g;
            // opaque to the optimiser: kept as "region base (scalar) + lane offset" it re-adds the base in front of every read
            asm volatile("" : "+v"(r0), "+v"(r1), "+v"(tp0), "+v"(tp1), "+v"(tp2), "+v"(tp3), "+v"(sp));
            // Software pipeline over the blocks, inside ONE wave (its instructions issue in order: S/dP -> exponentials -> dV/dK back to back
            // left the matrix pipe idle during the exponentials and the vector ALU idle during the products, 37 % of the sweep in issue stalls):
            // iteration ub runs the exponentials of block ub BETWEEN the S / dP products of block ub + 1 (one MFMA, four VALU, ...), then
            // requests the rows of block ub + 2 and issues dV / dK of block ub.  At the top of iteration ub: s, dp = S, dP − δ of block ub;
            // ls = lse of block ub; qf / gf = rows of block ub + 1 and dl = −δ of block ub + 1 (when it exists).
            // A block past the end (odd LT: the second query tile of the last block) reads the bytes behind the region — finite, never used.
            uint4 qf[2][2], gf[2][2];
            f32x4 ls[2], dl[2], s[2], dp[2];
            auto load_rows = [&](int ahead) {
                qf[0][0] = lds_u4(r0 + ahead);        gf[0][0] = lds_u4(r0 + ahead + 2048);
                qf[0][1] = lds_u4(r1 + ahead);        gf[0][1] = lds_u4(r1 + ahead + 2048);
                qf[1][0] = lds_u4(r0 + ahead + 4096); gf[1][0] = lds_u4(r0 + ahead + 6144);
                qf[1][1] = lds_u4(r1 + ahead + 4096); gf[1][1] = lds_u4(r1 + ahead + 6144);
            };
            auto load_ls = [&](int ahead) { ls[0] = lds_f4(sp + ahead); ls[1] = lds_f4(sp + ahead + 128); };
            auto load_dl = [&](int ahead) { dl[0] = lds_f4(sp + ahead + 64); dl[1] = lds_f4(sp + ahead + 192); };
            auto products = [&](int hq) {              // S and dP − δ of query tile hq of the block whose rows are in qf / gf
                s[hq] = mma(qf[hq][0], kf[0], zero4);   dp[hq] = mma(gf[hq][0], vf[0], dl[hq]);
                s[hq] = mma(qf[hq][1], kf[1], s[hq]);   dp[hq] = mma(gf[hq][1], vf[1], dp[hq]);
            };
            // KIND: 0 = every (query, key) of the block valid, 1 = per-element tests, 2 = the block's second query tile does not exist
            auto key_block = [&](auto kind_c, auto next_c, int ub) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr bool HASNEXT = decltype(next_c)::value;
                constexpr int HI = KIND == 2 ? 0 : 4096;
                bf16x8 gT[4], qT[4];
                qT[0] = tr_pair(tp0, tp0 + HI); gT[0] = tr_pair(tp0 + 2048, tp0 + 2048 + HI);
                qT[1] = tr_pair(tp1, tp1 + HI); gT[1] = tr_pair(tp1 + 2048, tp1 + 2048 + HI);
                qT[2] = tr_pair(tp2, tp2 + HI); gT[2] = tr_pair(tp2 + 2048, tp2 + 2048 + HI);
                qT[3] = tr_pair(tp3, tp3 + HI); gT[3] = tr_pair(tp3 + 2048, tp3 + 2048 + HI);
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 pf, sf;
                auto softmax = [&](int hq) {            // P and dS of query tile hq from s / dp (in place: the tile's accumulators die here)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(fmaf(s[hq][r], sc, -ls[hq][r]));
                        float dsv = pv * dp[hq][r];
                        if constexpr (KIND != 0) {
                            const int qrow = 32 * ub + 16 * hq + 4 * g + r;
                            const bool ok = qrow < L && key_ok && (!causal || key <= qrow);
                            pv = ok ? pv : 0.f;
                            dsv = ok ? dsv : 0.f;
                        }
                        pf[4 * hq + r] = (bf16_t)pv;
                        sf[4 * hq + r] = (bf16_t)dsv;
                    }
                };
                softmax(0);
                if constexpr (HASNEXT) products(0);                    // next block, first query tile: between the second tile's exponentials
                if constexpr (KIND == 2) {
#pragma unroll
                    for (int e = 4; e < 8; ++e) { pf[e] = (bf16_t)0.f; sf[e] = (bf16_t)0.f; }
                } else {
                    softmax(1);
                }
                if constexpr (HASNEXT) {
                    products(1);
                    __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? 16 : 28, 0);     // tile 0's exponentials
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA of the next block's first tile
                        __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? 4 : 7, 0);   // a quarter of tile 1's exponentials
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HASNEXT) {
                    // rows of block ub + 2 (of block ub + 1 again when there is none: in bounds, never used), −δ with them; lse of block ub + 1
                    const int more = ub + 2 < NP ? 1 : 0;
                    load_ls(256);
                    load_rows(8192 + 8192 * more);
                    load_dl(256 + 256 * more);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dVt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gT[dt], pf, dVt[dt], 0, 0, 0);
                    dKt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT[dt], sf, dKt[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                r0 += 8192; r1 += 8192; tp0 += 8192; tp1 += 8192; tp2 += 8192; tp3 += 8192; sp += 256;
            };
            if (ub_lo < NP) {
                const int last = NP - 1, nfull = LT >> 1;               // nfull = blocks with both query tiles
                // blocks [f0, f1) need no mask: whole query rows (32ub + 32 <= L), a whole key tile, and below the causal diagonal
                int f0 = causal ? (16 * kt + 46) >> 5 : 0, f1 = L >> 5;
                f0 = f0 < ub_lo ? ub_lo : f0;
                f1 = f1 > nfull ? nfull : f1;
                if (16 * kt + 16 > klen || f0 > f1) { f0 = nfull; f1 = nfull; }
                load_rows(0); load_ls(0); load_dl(0);
                products(0); products(1);
                __builtin_amdgcn_sched_barrier(0);
                if (ub_lo + 1 < NP) { load_rows(8192); load_dl(256); }
                int ub = ub_lo;
#pragma unroll 1
                for (; ub < f0 && ub < last; ++ub) key_block(cint<1>{}, ctrue{}, ub);
#ifdef ABWD_STAMPS
                const unsigned long long st_l0 = STAMP(); const int ub_s = ub;
#endif
#pragma unroll 1
                for (; ub < f1 && ub < last; ++ub) key_block(cint<0>{}, ctrue{}, ub);
#ifdef ABWD_STAMPS
                st_loop += STAMP() - st_l0; st_iter += ub - ub_s;
#endif
#pragma unroll 1
                for (; ub < last; ++ub) key_block(cint<1>{}, ctrue{}, ub);
                if (LT & 1) key_block(cint<2>{}, cfalse{}, last);
                else if (last >= f0 && last < f1) key_block(cint<0>{}, cfalse{}, last);
                else key_block(cint<1>{}, cfalse{}, last);
            }
            if (key < L) {          // lane owns key 16kt + li, d = 16dt + 4g + r
                bf16_t* krow = (bf16_t*)p.dk + (row0 + key) * p.ld_dqkv + (size_t)h * 64 + 4 * g;
                bf16_t* vrow = (bf16_t*)p.dv + (row0 + key) * p.ld_dqkv + (size_t)h * 64 + 4 * g;
                const size_t kbo = ((size_t)(2 * h) * (size_t)p.dqkv_kb_rows + row0 + key) * 32 + 4 * g, kbp = (size_t)p.dqkv_kb_rows * 32;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    store4(p.dqkv_kb_rows ? (bf16_t*)p.dk + kbo + (dt >> 1) * kbp + 16 * (dt & 1) : krow + 16 * dt, dKt[dt] * p.scale);
                    store4(p.dqkv_kb_rows ? (bf16_t*)p.dv + kbo + (dt >> 1) * kbp + 16 * (dt & 1) : vrow + 16 * dt, dVt[dt]);
                }
            }
        } else {
            // =============================== QRY unit: query tile qt, sweep over 32-key blocks (same pipeline)
            const int qt = unit - LT;
            const int qrow = 16 * qt + li;
            uint4 qf[2], gf[2];
            qf[0] = lds_u4(QG + qt * 4096 + offR0);        qf[1] = lds_u4(QG + qt * 4096 + offR1);
            gf[0] = lds_u4(QG + qt * 4096 + 2048 + offR0); gf[1] = lds_u4(QG + qt * 4096 + 2048 + offR1);
            const float lq = lds_f1(ST + qt * 128 + 4 * li);
            const float nd = lds_f1(ST + qt * 128 + 64 + 4 * li);
            const f32x4 nd4 = f32x4{nd, nd, nd, nd};
            f32x4 dQt[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dQt[dt] = zero4;
            int kend = klen;
            if (causal && 16 * qt + 16 < kend) kend = 16 * qt + 16;
            const int KB = (kend + 31) >> 5;                            // key blocks that hold a valid key (>= 1)
            lptr r0 = KV + offR0;
            lptr r1 = KV + offR1;
            lptr tp0 = KV + toff[0];
            lptr tp1 = KV + toff[1];
            lptr tp2 = KV + toff[2];
            lptr tp3 = KV + toff[3];
            asm volatile("" : "+v"(r0), "+v"(r1), "+v"(tp0), "+v"(tp1), "+v"(tp2), "+v"(tp3));
            const char* vp = vb + (size_t)li * rs + g * 16;            // VLDS = false: V rows of the block from global memory
            uint4 ka[2][2], va[2][2];
            f32x4 s[2], dp[2];
            auto load_kv = [&](int ahead_blocks, int kbk) {
                const int ahead = ahead_blocks * 2 * KVS;
                ka[0][0] = lds_u4(r0 + ahead); ka[0][1] = lds_u4(r1 + ahead);
                ka[1][0] = lds_u4(r0 + ahead + KVS); ka[1][1] = lds_u4(r1 + ahead + KVS);
                if (VLDS) {
                    va[0][0] = lds_u4(r0 + ahead + 2048); va[0][1] = lds_u4(r1 + ahead + 2048);
                    va[1][0] = lds_u4(r0 + ahead + KVS + 2048); va[1][1] = lds_u4(r1 + ahead + KVS + 2048);
                } else {
                    const int kb0 = 32 * (kbk + ahead_blocks);
                    if (kb0 + 32 <= L) {
                        const char* v0 = vp + (size_t)kb0 * rs;
                        va[0][0] = *(const uint4*)v0; va[0][1] = *(const uint4*)(v0 + 64);
                        va[1][0] = *(const uint4*)(v0 + 16 * rs); va[1][1] = *(const uint4*)(v0 + 16 * rs + 64);
                    } else {
                        int k0 = kb0 + li, k1 = k0 + 16;
                        k0 = k0 < L ? k0 : L - 1; k1 = k1 < L ? k1 : L - 1;
                        const char* v0 = vb + (size_t)k0 * rs + g * 16;
                        const char* v1 = vb + (size_t)k1 * rs + g * 16;
                        va[0][0] = *(const uint4*)v0; va[0][1] = *(const uint4*)(v0 + 64);
                        va[1][0] = *(const uint4*)v1; va[1][1] = *(const uint4*)(v1 + 64);
                    }
                }
            };
            auto products = [&](int hk) {              // S' and dP' − δ of key tile hk of the block whose rows are in ka / va
                s[hk] = mma(ka[hk][0], qf[0], zero4);   dp[hk] = mma(va[hk][0], gf[0], nd4);
                s[hk] = mma(ka[hk][1], qf[1], s[hk]);   dp[hk] = mma(va[hk][1], gf[1], dp[hk]);
            };
            auto qry_block = [&](auto kind_c, auto next_c, int kbk) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr bool HASNEXT = decltype(next_c)::value;
                constexpr int HI = KIND == 2 ? 0 : KVS;
                bf16x8 kT[4];
                kT[0] = tr_pair(tp0, tp0 + HI); kT[1] = tr_pair(tp1, tp1 + HI);
                kT[2] = tr_pair(tp2, tp2 + HI); kT[3] = tr_pair(tp3, tp3 + HI);
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 sf;
                auto softmax = [&](int hk) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[hk][r], sc, -lq));
                        float dsv = pv * dp[hk][r];
                        if constexpr (KIND != 0) {
                            const int keyv = 32 * kbk + 16 * hk + 4 * g + r;
                            const bool ok = keyv < klen && (!causal || keyv <= qrow);
                            dsv = ok ? dsv : 0.f;
                        }
                        sf[4 * hk + r] = (bf16_t)dsv;
                    }
                };
                softmax(0);
                if constexpr (HASNEXT) products(0);
                if constexpr (KIND == 2) {
#pragma unroll
                    for (int e = 4; e < 8; ++e) sf[e] = (bf16_t)0.f;
                } else {
                    softmax(1);
                }
                if constexpr (HASNEXT) {
                    products(1);
                    __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? 14 : 22, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? 4 : 6, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HASNEXT) load_kv(kbk + 2 < KB ? 2 : 1, kbk);         // block kbk + 2 (kbk + 1 again when there is none)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dQt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[dt], sf, dQt[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                r0 += 2 * KVS; r1 += 2 * KVS; tp0 += 2 * KVS; tp1 += 2 * KVS; tp2 += 2 * KVS; tp3 += 2 * KVS;
            };
            {
                const int last = KB - 1;
                const bool tail_last = 2 * last + 1 >= LT;              // the last swept block's second key tile does not exist
                // key blocks [0, f1) need no mask: all 32 keys valid and at or below the tile's first query under the causal mask
                int f1 = causal ? (16 * qt + 1) >> 5 : klen >> 5;
                f1 = f1 > (klen >> 5) ? klen >> 5 : f1;
                f1 = f1 > (LT >> 1) ? LT >> 1 : f1;
                load_kv(0, 0);
                products(0); products(1);
                __builtin_amdgcn_sched_barrier(0);
                if (1 < KB) load_kv(1, 0);
                int kbk = 0;
#pragma unroll 1
                for (; kbk < f1 && kbk < last; ++kbk) qry_block(cint<0>{}, ctrue{}, kbk);
#pragma unroll 1
                for (; kbk < last; ++kbk) qry_block(cint<1>{}, ctrue{}, kbk);
                if (tail_last) qry_block(cint<2>{}, cfalse{}, last);
                else if (last < f1) qry_block(cint<0>{}, cfalse{}, last);
                else qry_block(cint<1>{}, cfalse{}, last);
            }
            if (qrow < L) {         // lane owns query 16qt + li, d = 16dt + 4g + r
                bf16_t* drow = (bf16_t*)p.dq + (row0 + qrow) * p.ld_dqkv + (size_t)h * 64 + 4 * g;
                const size_t kbo = ((size_t)(2 * h) * (size_t)p.dqkv_kb_rows + row0 + qrow) * 32 + 4 * g, kbp = (size_t)p.dqkv_kb_rows * 32;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    store4(p.dqkv_kb_rows ? (bf16_t*)p.dq + kbo + (dt >> 1) * kbp + 16 * (dt & 1) : drow + 16 * dt, dQt[dt] * p.scale);
            }
        }
#ifdef ABWD_STAMPS
        st_u[unit < LT ? 0 : 1] += STAMP() - st_u0; st_n[unit < LT ? 0 : 1] += 1;
#endif
    }
#ifdef ABWD_STAMPS
    if (blockIdx.x == 1500 && lane == 0 && wave < 8) {
        unsigned long long* o = uia_abwd_stamps + wave * 8;
        o[0] = st_pro; o[1] = st_u[0]; o[2] = st_u[1]; o[3] = st_n[0]; o[4] = st_n[1]; o[5] = STAMP() - st_begin; o[6] = st_loop; o[7] = st_iter;
    }
#endif
}
// + slack behind the statistics: the pipeline requests one block ahead without asking whether its second tile exists, so the bytes of one
// (non-existent) K / V tile behind the K region and of a few statistics rows must lie inside the allocation (they are never used)
__host__ __device__ constexpr int units_lds_bytes(int LT, bool vlds) {
    const int kvs = vlds ? 4096 : 2048, need = kvs + 64 - LT * 128;
    return LT * 4096 + LT * kvs + LT * 128 + 16 + (need > 1024 ? need : 1024);
}


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

template <int NW, bool VLDS>
int launch_units(hipStream_t stream, const UiaAttnParams& p) {
    const int LT = (p.L + 15) / 16;
    auto kern = attn_bwd_units_kernel<NW, VLDS>;
    static UiaDevOnce attr_once;
    static_assert(units_lds_bytes(18, true) <= 160 * 1024, "LDS budget");
    UIA_ENSURE_LDS_ATTR(attr_once, kern, units_lds_bytes(18, VLDS));
    hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(64 * NW), units_lds_bytes(LT, VLDS), stream, p);
    UIA_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int uia_attn_bwd_launch(hipStream_t stream, int dtype, const UiaAttnParams& p, int cfg) {
    UIA_CHECK_ARG(cfg >= 0 && cfg <= 4, "uia_attn_bwd: unknown kernel configuration %d", cfg);
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
    if (cfg == 0) cfg = 2;
    if (cfg == 2) return launch_units<8, true>(stream, p);
    if (cfg == 3) return launch_units<4, false>(stream, p);
    if (cfg == 4) return launch_units<8, false>(stream, p);
    if (LT <= 8) return launch_bf16<8>(stream, p);
    if (LT <= 16) return launch_bf16<16>(stream, p);
    return launch_bf16<18>(stream, p);
}
