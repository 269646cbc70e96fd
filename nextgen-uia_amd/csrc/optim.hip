// optim.hip — one optimiser update over the FLAT fp32 adapter-parameter buffer:
//   global-norm gradient clipping + AdamW, the buffer that is also all-reduced (uia_comm).
//
// Reference semantics: /root/reference/src/models/biomedclip/finetune.py:297-302
//   torch.nn.utils.clip_grad_norm_(max_norm)  → coef = min(1, max_norm / (‖g‖₂ + 1e-6))
//   torch.optim.AdamW(lr, betas, eps=1e-8, weight_decay) (:244-249), decoupled decay p ← p·(1 − lr·wd),
//   bias-corrected moments, denom = sqrt(v̂) + eps.
// Two HBM-bound kernels (sum of squares → atomics; fused clip+AdamW); grads may be pre-scaled
// (grad_scale = 1/world after an all-reduce SUM).
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(size_t n, const float* __restrict__ g, float gscale, float* __restrict__ acc) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i] * gscale;
        s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, float lr, float b1, float b2, float eps, float wd, float max_norm,
                                                     float bc1, float bc2s, float gscale, const float* __restrict__ sumsq) {
    float coef = gscale;
    if (max_norm > 0.f) {
        const float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
        coef *= c < 1.0f ? c : 1.0f;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = fmaf(b1, m[i], (1.0f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

}  // namespace

int uia_adamw_clip_launch(hipStream_t stream, size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float max_norm, int step, float grad_scale, float* ws) {
    UIA_CHECK_ARG(n > 0 && p && g && m && v && ws && step >= 1, "uia_adamw_clip_step: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    size_t blocks = (n + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    UIA_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(float), stream));
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, g, grad_scale, ws);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, max_norm,
                       bc1, bc2s, grad_scale, ws);
    UIA_CHECK_LAUNCH();
    return 0;
}
