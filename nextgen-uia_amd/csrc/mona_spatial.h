// mona_spatial.h — device helpers shared by the Mona spatial kernels (mona.hip) and the fused forward adapter (mona_fused.hip):
// the noise estimator's forward, the merged 7x7 stencil tap, the row-strip stencil and the LDS tile geometry of the bf16 fast path.
// Reference: /root/reference/src/adapters/mona.py:75-93,159-195,261-295,370-424 (the four *MonaOp.forward bodies).
#pragma once
#include "uia_common.h"
#include "uia_kernels.h"

namespace uia_mona {

constexpr int BOTT = 64;

// merged-tap weight of stencil position (i,j) in the 7×7 frame
#define KM(i, j) (w3 * k3[(i) * 7 + (j)] + (((i) >= 1 && (i) <= 5 && (j) >= 1 && (j) <= 5) ? w2 * k2[((i) - 1) * 5 + ((j) - 1)] : 0.f) + \
                  (((i) >= 2 && (i) <= 4 && (j) >= 2 && (j) <= 4) ? w1 * k1[((i) - 2) * 3 + ((j) - 2)] : 0.f))

template <typename T>
__device__ __forceinline__ void load_tokens(const T* __restrict__ src, float* __restrict__ dst, int ntok, int tid) {
    for (int i = tid; i < ntok * 8; i += 512) {
        float v[8];
        load8(src + (size_t)i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[i * 8 + e] = v[e];
    }
}

// Noise estimator forward (mona.py:170-176,187 / :393-399,416): pool → 1×1 (64→16) → ReLU → 1×1 (16→3) → softmax.
// scr: ≥ 4*64 + 64 + 16 + 16 + 4 floats.  Returns w1,w2,w3 in scr[W_OFF..].
constexpr int NGRP = 8;                   // pixel groups = waves per image (512 threads)
constexpr int SCR_PART = 0, SCR_POOL = 512, SCR_HPRE = 576, SCR_HID = 592, SCR_W = 608, SCR_DPOOL = 616, SCR_RED = 680, SCR_SIZE = 720;
constexpr int RED_FLOATS = 64 * 50;
// layout of one per-image partial-gradient row in the workspace (floats)
constexpr int WS_PROJ_W = 0, WS_PROJ_B = 4096, WS_C1W = 4160, WS_C1B = 4736, WS_C2W = 4800, WS_C2B = 6400, WS_C3W = 6464, WS_C3B = 9600,
              WS_FREQ = 9664, WS_NE1W = 9728, WS_NE1B = 10752, WS_NE3W = 10768, WS_NE3B = 10816, WS_ROW = 10880;       // pass-A per-channel accumulators (LDS atomics from the 8 pixel groups)

__device__ __forceinline__ void noise_forward(const uia_mona_spatial_desc& p, const float* tS, float f, int hw, int ppg, int c, int grp,
                                              int tid, float* scr) {
    float s = 0.f;
    const int p0 = grp * ppg, p1 = min(hw, p0 + ppg);
    for (int px = p0; px < p1; ++px) s += tS[(1 + px) * BOTT + c];
    scr[SCR_PART + grp * 64 + c] = s;
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < NGRP; ++q) a += scr[SCR_PART + q * 64 + tid];
        scr[SCR_POOL + tid] = f * a / hw;
    }
    __syncthreads();
    if (tid < 16) {
        float a = p.ne1_b[tid];
        for (int k = 0; k < 64; ++k) a = fmaf(p.ne1_w[tid * 64 + k], scr[SCR_POOL + k], a);
        scr[SCR_HPRE + tid] = a;
        scr[SCR_HID + tid] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (tid == 0) {
        float l[3];
        for (int k = 0; k < 3; ++k) {
            float a = p.ne3_b[k];
            for (int j = 0; j < 16; ++j) a = fmaf(p.ne3_w[k * 16 + j], scr[SCR_HID + j], a);
            l[k] = a;
        }
        const float m = fmaxf(l[0], fmaxf(l[1], l[2]));
        const float e0 = expf(l[0] - m), e1 = expf(l[1] - m), e2 = expf(l[2] - m), inv = 1.f / (e0 + e1 + e2);
        scr[SCR_W + 0] = e0 * inv; scr[SCR_W + 1] = e1 * inv; scr[SCR_W + 2] = e2 * inv;
    }
    __syncthreads();
}

constexpr int FLD = 68;
constexpr int FAST_RED = 4 * 64 * 51;                          // four [64 channels][49 taps + Σdc, stride 51] reduction images
constexpr int KW_FLOATS = 64 * (9 + 25 + 49);                  // stencil-weight staging area (forward: its own; backward: inside the dz tile)
__host__ __device__ constexpr size_t spatial_fast_lds(int hw, bool bwd) {
    return ((size_t)(hw + 1) * BOTT + (size_t)((hw * FLD > FAST_RED) ? hw * FLD : FAST_RED) +
            (bwd ? (size_t)(((hw + 1) * FLD > KW_FLOATS) ? (hw + 1) * FLD : KW_FLOATS) : (size_t)KW_FLOATS) + SCR_SIZE) * sizeof(float);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    return r;
}

// acc[x] += Σ_j km[i][j] · src(row yy)[x + (j-3)·sgn]  for the 7 source rows of output row y.  FLIP = transposed stencil.
template <int W, bool FLIP>
__device__ __forceinline__ void stencil_row(const float* __restrict__ src, int ld, int h, int y, int c, const float (&km)[49], float (&acc)[W]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int yy = FLIP ? y - (i - 3) : y + (i - 3);
        if (yy < 0 || yy >= h) continue;                       // wave-uniform
        float row[W + 6];
#pragma unroll
        for (int x = 0; x < W + 6; ++x) row[x] = 0.f;
#pragma unroll
        for (int x = 0; x < W; ++x) row[3 + x] = src[(size_t)(yy * W + x) * ld + c];
#pragma unroll
        for (int x = 0; x < W; ++x)
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[x] = fmaf(km[i * 7 + j], row[3 + x + (FLIP ? -(j - 3) : (j - 3))], acc[x]);
    }
}


}  // namespace uia_mona
