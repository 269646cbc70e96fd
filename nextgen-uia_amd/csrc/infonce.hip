// infonce.hip — symmetric InfoNCE loss, forward + gradient w.r.t. both feature matrices, fp32.
//
// Reference: /root/reference/src/losses/losses.py:23-47 (F.normalize eps 1e-12, logits = Î·T̂ᵀ/τ,
// cross-entropy both ways against the diagonal, mean of the two).  Backward: SURVEY.md Appendix E.3.
// 67 MFLOP at B=256, E=512: not worth an MFMA path; five small fp32 VALU kernels, everything stays
// L2-resident.  Workspace (floats): 2·B·E (Î, T̂) + 2·B (norms) + B·B (S, then dS) + 2·B (row / column lse).
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

__global__ __launch_bounds__(256) void normalize_kernel(int B, int E, const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ an, float* __restrict__ bn, float* __restrict__ norms) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= 2 * B) return;
    const float* src = row < B ? a + (size_t)row * E : b + (size_t)(row - B) * E;
    float* dst = row < B ? an + (size_t)row * E : bn + (size_t)(row - B) * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(src[e], src[e], s);
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    const float inv = 1.0f / nrm;
    for (int e = lane; e < E; e += 64) dst[e] = src[e] * inv;
    if (lane == 0) norms[row] = nrm;
}

// S[i][j] = <an_i, bn_j> * inv_temp ; 16x16 outputs per block
__global__ __launch_bounds__(256) void logits_kernel(int B, int E, const float* __restrict__ an, const float* __restrict__ bn, float inv_temp,
                                                      float* __restrict__ S) {
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4), j = blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= B || j >= B) return;
    const float* x = an + (size_t)i * E;
    const float* y = bn + (size_t)j * E;
    float s = 0.f;
    for (int e = 0; e < E; e += 4) {
        const f32x4 u = load4(x + e), v = load4(y + e);
        s = fmaf(u[0], v[0], s); s = fmaf(u[1], v[1], s); s = fmaf(u[2], v[2], s); s = fmaf(u[3], v[3], s);
    }
    S[(size_t)i * B + j] = s * inv_temp;
}

// wave r < B: row r; wave r >= B: column r-B.  loss += (lse - diag) / (2B)
__global__ __launch_bounds__(256) void lse_kernel(int B, const float* __restrict__ S, float* __restrict__ lse_r, float* __restrict__ lse_c,
                                                   float* __restrict__ loss) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= 2 * B) return;
    const bool col = r >= B;
    const int k = col ? r - B : r;
    const size_t base = col ? (size_t)k : (size_t)k * B, stride = col ? (size_t)B : 1;
    float m = -INFINITY;
    for (int t = lane; t < B; t += 64) m = fmaxf(m, S[base + t * stride]);
    m = wave_max(m);
    float s = 0.f;
    for (int t = lane; t < B; t += 64) s += expf(S[base + t * stride] - m);
    const float l = m + logf(wave_sum(s));
    if (lane == 0) {
        (col ? lse_c : lse_r)[k] = l;
        atomicAdd(loss, (l - S[(size_t)k * B + k]) / (2.0f * B));
    }
}

__global__ void dlogits_kernel(int B, float gscale, float* __restrict__ S, const float* __restrict__ lse_r, const float* __restrict__ lse_c) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * B) return;
    const int i = idx / B, j = idx - i * B;
    const float s = S[idx];
    S[idx] = (expf(s - lse_r[i]) + expf(s - lse_c[j]) - (i == j ? 2.0f : 0.0f)) * gscale / (2.0f * B);
}

// one block per output row r (r < B: image row, else text row): dn = inv_temp · Σ_j dS(i,j)·other_j ;
// d = (dn − n̂·<n̂,dn>) / ‖·‖
__global__ __launch_bounds__(256) void dfeat_kernel(int B, int E, const float* __restrict__ dS, const float* __restrict__ an, const float* __restrict__ bn,
                                                     const float* __restrict__ norms, float inv_temp, float* __restrict__ da, float* __restrict__ db) {
    __shared__ float red[256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const bool txt = r >= B;
    const int i = txt ? r - B : r;
    const float* other = txt ? an : bn;
    const float* self = (txt ? bn : an) + (size_t)i * E;
    float* out = (txt ? db : da) + (size_t)i * E;
    // a thread owns four consecutive features (16-byte loads of `other`) and every second j: 128 threads cover a 512-wide row, the two
    // halves of the block split the sweep over j and meet in LDS.  (One feature per thread and the whole sweep per thread: 135 us at
    // B = 256 with 512 KB of L2 reads per block behind 4-byte loads.)
    constexpr int MAXQ = 2;                       // E ≤ 1024
    const int half = tid >> 7, q = tid & 127;
    f32x4 acc[MAXQ];
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    __shared__ float wS[256];
    __shared__ f32x4 part[MAXQ][128];
    for (int j0 = 0; j0 < B; j0 += 256) {
        __syncthreads();
        if (j0 + tid < B) wS[tid] = txt ? dS[(size_t)(j0 + tid) * B + i] : dS[(size_t)i * B + j0 + tid];
        __syncthreads();
        const int jn = B - j0 < 256 ? B - j0 : 256;
#pragma unroll 8
        for (int j = half; j < jn; j += 2) {
            const float w = wS[j];
            const f32x4* row = (const f32x4*)(other + (size_t)(j0 + j) * E);
#pragma unroll
            for (int k = 0; k < MAXQ; ++k) {
                const int e4 = q + 128 * k;
                if (4 * e4 < E) {
                    const f32x4 o = row[e4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[k][c] = fmaf(w, o[c], acc[k][c]);
                }
            }
        }
    }
    if (half == 1) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) part[k][q] = acc[k];
    }
    __syncthreads();
    float dot = 0.f;
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int e4 = q + 128 * k;
            acc[k] = (acc[k] + part[k][q]) * inv_temp;
            if (4 * e4 < E) {
                const f32x4 sv = *(const f32x4*)(self + 4 * e4);
#pragma unroll
                for (int c = 0; c < 4; ++c) dot = fmaf(acc[k][c], sv[c], dot);
            }
        }
    }
    red[tid] = dot;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    dot = red[0];
    const float inv = 1.0f / norms[r];
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int e4 = q + 128 * k;
            if (4 * e4 < E) {
                const f32x4 sv = *(const f32x4*)(self + 4 * e4);
                *(f32x4*)(out + 4 * e4) = (acc[k] - sv * dot) * inv;
            }
        }
    }
}

}  // namespace

size_t uia_infonce_workspace_floats(int B, int E) { return (size_t)2 * B * E + 2 * B + (size_t)B * B + 2 * B; }

int uia_infonce_launch(hipStream_t stream, int B, int E, const float* img, const float* txt, float inv_temp, float grad_scale, float* loss,
                       float* dimg, float* dtxt, float* ws, size_t ws_floats) {
    UIA_CHECK_ARG(B > 0 && E > 0 && E % 4 == 0 && E <= 1024, "uia_infonce: unsupported shape B=%d E=%d", B, E);
    UIA_CHECK_ARG(img && txt && loss && ws, "uia_infonce: null tensor");
    UIA_CHECK_ARG((dimg == nullptr) == (dtxt == nullptr), "uia_infonce: pass both gradient buffers or neither");
    UIA_CHECK_ARG(ws_floats >= uia_infonce_workspace_floats(B, E), "uia_infonce: workspace too small");
    UIA_CHECK_ARG(((uintptr_t)ws | (uintptr_t)dimg | (uintptr_t)dtxt) % 16 == 0, "uia_infonce: workspace and gradient buffers must be 16-byte aligned");
    float* an = ws;
    float* bn = an + (size_t)B * E;
    float* norms = bn + (size_t)B * E;
    float* S = norms + 2 * B;
    float* lse_r = S + (size_t)B * B;
    float* lse_c = lse_r + B;
    UIA_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), stream));
    hipLaunchKernelGGL(normalize_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, stream, B, E, img, txt, an, bn, norms);
    hipLaunchKernelGGL(logits_kernel, dim3((B + 15) / 16, (B + 15) / 16), dim3(256), 0, stream, B, E, an, bn, inv_temp, S);
    hipLaunchKernelGGL(lse_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, stream, B, S, lse_r, lse_c, loss);
    if (dimg) {
        hipLaunchKernelGGL(dlogits_kernel, dim3((B * B + 255) / 256), dim3(256), 0, stream, B, grad_scale, S, lse_r, lse_c);
        hipLaunchKernelGGL(dfeat_kernel, dim3(2 * B), dim3(256), 0, stream, B, E, S, an, bn, norms, inv_temp, dimg, dtxt);
    }
    UIA_CHECK_LAUNCH();
    return 0;
}
