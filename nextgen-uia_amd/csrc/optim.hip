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

// ---- guarded forms (round 5): the training loops run WITHOUT a host read of the loss.  The reference decides on the host
// (finetune.py:281-285 `if not torch.isfinite(loss): continue` — the micro-batch's backward AND the update check of that
// iteration are skipped); here the same two decisions are taken on the device from the loss the InfoNCE kernel left in HBM:
//   accum:  the micro-batch's gradients land in a STAGING buffer; they are added to the cycle's accumulator only when the
//           loss is finite, the staging buffer is zeroed either way, the flag of this micro-batch is left in acc[n]
//           (0 = finite, 1 = not), which travels through the data-parallel all-reduce with the gradients;
//   update: runs only when acc[n] == 0 (summed over ranks: the boundary micro-batch was finite on every rank); the
//           update counter t, the cosine learning rate of update t and the bias corrections live on the device, so a
//           skipped update does not advance the schedule — exactly what the host-side `continue` did.
__global__ __launch_bounds__(256) void accum_guarded_kernel(size_t n, float* __restrict__ acc, float* __restrict__ mb, const float* __restrict__ loss,
                                                            float* __restrict__ stats, int* __restrict__ ctl, unsigned char* __restrict__ ok_log, long log_index) {
    const float l = *loss;
    const bool ok = __builtin_isfinite(l);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = mb[i];
        if (ok) acc[i] += v;
        mb[i] = 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc[n] = ok ? 0.0f : 1.0f;
        if (ok) {
            stats[0] += l;                                      // sum of the finite losses since the host last cleared it
            ctl[1] += 1;
        } else {
            ctl[2] += 1;
        }
        if (ok_log) ok_log[log_index] = ok ? 1 : 0;
    }
}

// one thread: the decision and the scalars of this update.  ws: [0] squared norm (zeroed here), [1] run the update (1 / 0), [2] lr, [3] bc1, [4] sqrt(bc2)
__global__ void guarded_prepare_kernel(const float* __restrict__ acc_flag, int* __restrict__ ctl, float* __restrict__ ws, float lr_base, float lr_min,
                                       int t_max, float b1, float b2) {
    const bool run = acc_flag == nullptr || *acc_flag == 0.0f;
    ws[0] = 0.0f;
    ws[1] = run ? 1.0f : 0.0f;
    if (!run) {
        ctl[3] += 1;                                            // updates skipped on a non-finite boundary micro-batch
        return;
    }
    const int t = ctl[0];
    ctl[0] = t + 1;
    double lr = lr_base;
    if (t_max > 0) lr = (double)lr_min + ((double)lr_base - (double)lr_min) * (1.0 + cos(3.14159265358979323846 * (double)t / (double)t_max)) * 0.5;   // CosineAnnealingLR closed form
    ws[2] = (float)lr;
    ws[3] = 1.0f - powf(b1, (float)(t + 1));
    ws[4] = sqrtf(1.0f - powf(b2, (float)(t + 1)));
}

__global__ __launch_bounds__(256) void sumsq_guarded_kernel(size_t n, const float* __restrict__ g, float gscale, float* __restrict__ ws) {
    if (ws[1] == 0.0f) return;
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = g[i] * gscale;
        s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(ws, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_guarded_kernel(size_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                            float b1, float b2, float eps, float wd, float max_norm, float gscale, float skip_scale,
                                                            const float* __restrict__ ws) {
    if (ws[1] == 0.0f) {
        // skipped: the accumulator keeps its content and the cycle goes on (reference: no zero_grad without a step).  Data parallel: the buffer
        // holds the SUM over ranks on every rank now; scaled by 1/world it is this rank's share again, so that the next all-reduce restores it
        if (skip_scale != 1.0f)
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] *= skip_scale;
        return;
    }
    const float lr = ws[2], bc1 = ws[3], bc2s = ws[4];
    float coef = gscale;
    if (max_norm > 0.f) {
        const float c = max_norm / (sqrtf(ws[0]) + 1e-6f);
        coef *= c < 1.0f ? c : 1.0f;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        g[i] = 0.0f;                                            // optimizer.zero_grad() of the reference, in the same pass
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

int uia_grad_accum_guarded_launch(hipStream_t stream, size_t n, float* acc, float* mb, const float* loss, float* stats, int* ctl, unsigned char* ok_log, long log_index) {
    UIA_CHECK_ARG(n > 0 && acc && mb && loss && stats && ctl && (ok_log == nullptr || log_index >= 0), "uia_grad_accum_guarded: bad arguments");
    size_t blocks = (n + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(accum_guarded_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, acc, mb, loss, stats, ctl, ok_log, log_index);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_adamw_clip_guarded_launch(hipStream_t stream, size_t n, float* p, float* acc, float* m, float* v, float lr, float lr_min, int t_max, float beta1,
                                  float beta2, float eps, float weight_decay, float max_norm, float grad_scale, float skip_scale, float* ws8, int* ctl) {
    UIA_CHECK_ARG(n > 0 && p && acc && m && v && ws8 && ctl && t_max >= 0, "uia_adamw_clip_step_guarded: bad arguments");
    size_t blocks = (n + 255) / 256;
    blocks = blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(guarded_prepare_kernel, dim3(1), dim3(1), 0, stream, acc + n, ctl, ws8, lr, lr_min, t_max, beta1, beta2);
    hipLaunchKernelGGL(sumsq_guarded_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, acc, grad_scale, ws8);
    hipLaunchKernelGGL(adamw_guarded_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, p, acc, m, v, beta1, beta2, eps, weight_decay, max_norm, grad_scale, skip_scale, ws8);
    UIA_CHECK_LAUNCH();
    return 0;
}

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
