// uia_common.h — shared device/host helpers for libuia_hip.so (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, hard-coded.
//   * "T" is the activation/operand element type: __bf16 (UIA_BF16) or float (UIA_F32).
//     All arithmetic is fp32; T only decides how operands are stored in HBM/LDS.
//   * the residual stream and every parameter gradient are fp32 in both modes.
//   * all tensors are row-major contiguous unless a leading dimension is passed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define UIA_WAVE 64

// ---------------------------------------------------------------- error plumbing (host)
void uia_set_error(const char* fmt, ...);
#define UIA_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            uia_set_error(__VA_ARGS__);          \
            return -1;                           \
        }                                        \
    } while (0)
#define UIA_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            uia_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),         \
                          __FILE__, __LINE__);                                           \
            return -2;                                                                   \
        }                                                                                \
    } while (0)
#define UIA_CHECK_LAUNCH() UIA_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------- per-device launch attributes
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is recorded PER DEVICE, so the "already done" flag of a kernel
// instantiation is a bit per device ordinal (a process that touches a second GPU must set it there too), atomically
// updated (two host threads racing set the same idempotent attribute twice at worst).
#include <atomic>
struct UiaDevOnce {
    std::atomic<unsigned long long> mask{0};
};
#define UIA_ENSURE_LDS_ATTR(once, kern, bytes)                                                                     \
    do {                                                                                                           \
        int _dev = 0;                                                                                              \
        UIA_CHECK_HIP(hipGetDevice(&_dev));                                                                        \
        const unsigned long long _bit = 1ull << (_dev & 63);                                                       \
        if (!((once).mask.load(std::memory_order_acquire) & _bit)) {                                               \
            UIA_CHECK_HIP(hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
            (once).mask.fetch_or(_bit, std::memory_order_release);                                                 \
        }                                                                                                          \
    } while (0)

#if defined(__HIPCC__)
// Three-byte tensors (round 4; include/uia_hip.h, uia_gemm_desc.resid_lo8): a value is its bf16 hi plane + a signed low byte,
// float bits = (hi_bits << 16) + (lo << 8).  Inside a frozen block the attention-half output x1 and its gradient dx1 travel in that form
// between the GEMM epilogues and this kernel: 3 bytes read instead of 4, 3 written instead of 4 + 2.
__device__ __forceinline__ f32x4 three_byte_load4(const bf16_t* hi, const int8_t* lo) {
    typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
    const u16x4 h = *(const u16x4*)hi;
    const int l = *(const int*)lo;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(float, ((unsigned)h[e] << 16) + ((unsigned)__builtin_amdgcn_sbfe(l, 8 * e, 8) << 8));
    return r;
}
__device__ __forceinline__ void three_byte_store4(bf16_t* hi, int8_t* lo, f32x4 v) {
    unsigned w[4];
    unsigned short hs[4];
    const float f[4] = {v[0], v[1], v[2], v[3]};      // (bit-casting v[e] of the ext_vector inside the unrolled loop read element 0 for every e: hipcc 7.2)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned vb = __builtin_bit_cast(unsigned, f[e]);
        hs[e] = __builtin_bit_cast(unsigned short, (bf16_t)f[e]);
        int d = (int)((vb + 0x80u) >> 8) - (int)((unsigned)hs[e] << 8);
        d = d < -127 ? -127 : (d > 127 ? 127 : d);
        w[e] = (unsigned)d & 0xFFu;
    }
    uint2 hp;
    hp.x = (unsigned)hs[0] | ((unsigned)hs[1] << 16);
    hp.y = (unsigned)hs[2] | ((unsigned)hs[3] << 16);
    *(uint2*)hi = hp;
    *(unsigned*)lo = w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24);
}

__device__ __forceinline__ f32x4 three_byte_decode4(uint2 hraw, unsigned lraw) {
    const unsigned h[4] = {hraw.x & 0xFFFFu, hraw.x >> 16, hraw.y & 0xFFFFu, hraw.y >> 16};
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_bit_cast(float, (h[e] << 16) + ((unsigned)__builtin_amdgcn_sbfe((int)lraw, 8 * e, 8) << 8));
    return r;
}

#endif

// compute units of the current device (cached on first use; 256 on MI355X)
inline int uia_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
    }
    return n;
}

// ---------------------------------------------------------------- element load/store
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// 4 consecutive elements <-> f32x4
__device__ __forceinline__ f32x4 load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    bf16x4 v = *(const bf16x4*)p;
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4*)p = r;
}
// 8 consecutive elements (16 B of bf16 / 32 B of f32)
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
    bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *(f32x4*)p = a;
    *(f32x4*)(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    *(bf16x8*)p = r;
}

// ---------------------------------------------------------------- math
// Φ(x) (standard normal CDF) and E = exp(-x²/2) from ONE v_exp + ONE v_rcp (Abramowitz & Stegun 7.1.25/26):
//   FAST = false: 5-term form, |Δerf| ≤ 1.5e-7  (fp32 parity mode)
//   FAST = true : 3-term form, |Δerf| ≤ 2.5e-5 → |Δgelu| ≤ 2.6e-5, an order below bf16 rounding (bf16 mode)
// gelu(x) = x·Φ, gelu'(x) = Φ + x·E/√(2π): the backward reuses the same exponential.
template <bool FAST>
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& E) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    E = __expf(-ax * ax);
    float y;
    if (FAST) {
        const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f, ax, 1.0f));
        y = fmaf(fmaf(0.7478556f, t, -0.0958798f), t, 0.3480242f) * t;
    } else {
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
        y = fmaf(1.061405429f, t, -1.453152027f);
        y = fmaf(y, t, 1.421413741f);
        y = fmaf(y, t, -0.284496736f);
        y = fmaf(y, t, 0.254829592f) * t;
    }
    const float half_erfc = 0.5f * y * E;                 // 0.5·erfc(|x|/√2)
    Phi = x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
template <bool FAST = false> __device__ __forceinline__ float gelu_erf_t(float x) {
    float P, E;
    gelu_parts<FAST>(x, P, E);
    return x * P;
}
template <bool FAST = false> __device__ __forceinline__ float dgelu_erf_t(float x) {
    float P, E;
    gelu_parts<FAST>(x, P, E);
    return fmaf(x * 0.39894228040143268f, E, P);
}
// exact-erf GELU (timm / HF BERT / F.gelu default) and its derivative, accurate form
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_t<false>(x); }
__device__ __forceinline__ float dgelu_erf(float x) { return dgelu_erf_t<false>(x); }
// QuickGELU x*sigmoid(1.702x) (reference src/third_party/openai_clip/model.py:172-174)
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
}
__device__ __forceinline__ float dquick_gelu(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    return s * fmaf(1.702f * x, 1.0f - s, 1.0f);
}

// bf16-mode GELU for the GEMM epilogues, two values per instruction and no transcendental:
//   Φ(x)     ≈ ½ + xc·P(xc²)    gelu'(x) = Φ(x) + x·φ(x) ≈ ½ + xc·Q(xc²),    xc = clamp(x, -4, 4)
// P, Q are degree-7 minimax fits on [-4, 4] (tools/fit_gelu_poly.py): |ΔΦ| ≤ 5.3e-5, |Δgelu'| ≤ 5e-4 (incl. the clamp at 4),
// both an order below bf16 rounding (2^-9 = 2e-3).  The A&S form above costs a v_exp and a v_rcp per value (quarter-rate:
// 32 of its ≈90 issue cycles per wave); this is 8 packed FMAs per PAIR of values (v_pk_fma_f32), ≈4× cheaper, and the
// fc1/fc2-dgrad epilogues were VALU-bound on exactly that (45K cycles per 256×256 tile against 15K without GELU).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 uia_clamp4(f32x2 x) {
    f32x2 r = {__builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f)};
    return r;
}
__device__ __forceinline__ f32x2 uia_odd_poly8(f32x2 xc, const float (&c)[8]) {
    const f32x2 u = xc * xc;
    f32x2 a = {c[7], c[7]};
#pragma unroll
    for (int k = 6; k >= 0; --k) a = a * u + c[k];
    return xc * a + 0.5f;
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    const float c[8] = {3.988475093e-01f, -6.617537764e-02f, 9.664873136e-03f, -1.048204087e-03f, 8.066734772e-05f, -4.100862563e-06f, 1.217109478e-07f, -1.580783585e-09f};
    return x * uia_odd_poly8(uia_clamp4(x), c);
}
__device__ __forceinline__ f32x2 dgelu_poly2(f32x2 x) {
    const float c[8] = {7.967216367e-01f, -2.620298342e-01f, 5.591481834e-02f, -7.687439325e-03f, 6.876450573e-04f, -3.845953927e-05f, 1.213804812e-06f, -1.641976469e-08f};
    return uia_odd_poly8(uia_clamp4(x), c);
}

// ---------------------------------------------------------------- wave reductions
// On the VALU (round 4): __shfl_xor compiles to ds_bpermute_b32 — a trip through the LDS crossbar per step, six dependent trips per reduction, four
// reductions per row of the LayerNorm backward.  gfx950 can do the whole butterfly without LDS: quad_perm / row mirrors (DPP) inside a row of 16 lanes,
// v_permlane16_swap / v_permlane32_swap across rows.  Order: partners at distance 1, 2, then the other quad, the other half-row, the other row, the other
// half-wave — every lane ends with the same sum (the values inside a group are identical before the group meets its mirror image).
__device__ __forceinline__ float uia_dpp_quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float uia_dpp_quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float uia_dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }  // row_half_mirror
__device__ __forceinline__ float uia_dpp_row_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); }   // row_mirror
// rows_*: all-reduce over the four rows of 16 lanes (lanes li, li + 16, li + 32, li + 48): the other row of the pair, then the other pair.
// v_permlane16_swap a, b: rows 1 and 3 of a change places with rows 0 and 2 of b; v_permlane32_swap: the upper half of a with the lower half of b — with
// a = b = v the two registers then hold the two partners of every lane.  Inline asm: the builtin of this hipcc (7.2) returns the first register for BOTH
// results (tools/attic/dpp_test.hip shows it); the s_nop pairs are the wait states the hazard pass would have placed around a VALU lane permute.
__device__ __forceinline__ void uia_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void uia_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_sum(float v) {
    float a = v, b = v;
    uia_swap16(a, b);
    a += b;
    b = a;
    uia_swap32(a, b);
    return a + b;
}
__device__ __forceinline__ float rows_max(float v) {
    float a = v, b = v;
    uia_swap16(a, b);
    a = fmaxf(a, b);
    b = a;
    uia_swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef UIA_WAVE_REDUCE_LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    v += uia_dpp_quad_xor1(v);
    v += uia_dpp_quad_xor2(v);
    v += uia_dpp_half_mirror(v);
    v += uia_dpp_row_mirror(v);
    return rows_sum(v);
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef UIA_WAVE_REDUCE_LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
#else
    v = fmaxf(v, uia_dpp_quad_xor1(v));
    v = fmaxf(v, uia_dpp_quad_xor2(v));
    v = fmaxf(v, uia_dpp_half_mirror(v));
    v = fmaxf(v, uia_dpp_row_mirror(v));
    return rows_max(v);
#endif
}

// ---------------------------------------------------------------- counter-based dropout RNG
// keep(idx) is a pure function of (seed, idx): forward and backward regenerate the same mask,
// so no mask tensor is stored.  (The reference uses torch's Philox stream,
// src/adapters/mona.py:109,147; bit-matching that stream is a documented non-goal.)
__device__ __forceinline__ uint32_t uia_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// Eight consecutive elements per draw (the LoRA input dropout: uia_dropout, the N = 64 stream kernel's A operand, the run-time GEMM
// epilogue): bit e of the result = keep element 8·grp + e.  Two hashes per group instead of three per element — the stand-alone pass
// was bound by its integer arithmetic (34 us for 134 MB), and a fused mask must not cost more than the pass it replaces.
// thresh16 = round(p · 65536): the drop probability is quantised to 1/65536.
__device__ __forceinline__ uint32_t dropout_keep8(uint64_t seed, uint32_t grp, uint32_t thresh16) {
    // two full mixes of (group, seed), then three xorshift32 steps: v_mul_lo_u32 runs at a quarter of the integer rate, and with a full mix per
    // 32-bit word the mask cost 12 us of the 40 us x·Aᵀ launch that carries it
    uint32_t h = uia_hash32(uia_hash32(grp ^ (uint32_t)seed) + (uint32_t)(seed >> 32));
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        m |= ((h & 0xFFFFu) >= thresh16 ? 1u : 0u) << (2 * q);
        m |= ((h >> 16) >= thresh16 ? 1u : 0u) << (2 * q + 1);
        h ^= h << 13; h ^= h >> 17; h ^= h << 5;
    }
    return m;
}
__host__ __device__ inline uint32_t dropout_thresh16(float p) {
    const float t = p * 65536.0f + 0.5f;
    return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t idx, uint32_t thresh) {
    // thresh = floor(p * 2^32); keep iff hash >= thresh
    uint32_t h = uia_hash32(idx ^ (uint32_t)seed) ^ uia_hash32((idx * 0x9E3779B9U) + (uint32_t)(seed >> 32));
    return uia_hash32(h) >= thresh;
}
