// heads.hip — the two spatial ops of the FPN task heads (reference /root/reference/src/third_party/timm/clip_adapter.py:47-57):
//   * nn.Upsample((H,W), mode="bilinear", align_corners=False) of a token-major map (seg head), forward and backward;
//   * nn.AdaptiveAvgPool2d(1) + Flatten = mean over an image's tokens (cls head), forward and backward.
// Both are HBM-bound maps with a few taps per element; fp32 in both modes (they sit after the last GEMM of the head).
//
// The seg head of the reference upsamples 512 channels to 224×224 and then applies Conv1×1(512→classes).  A 1×1 convolution
// (a per-pixel channel mix plus a bias) commutes with bilinear interpolation, whose weights sum to one, so the host side
// applies the convolution on the 14×14 grid (a GEMM) and this kernel interpolates `classes` channels: same result,
// 256× less data (at B=256 the reference's intermediate is 26 GB).
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

// PyTorch area_pixel_compute_source_index(align_corners=false, cubic=false): src = max(0, scale·(dst+0.5) − 0.5)
__device__ __forceinline__ void src_taps(int dst, float scale, int in, int& i0, int& i1, float& l1) {
    float s = scale * (dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < in - 1 ? i0 : in - 1;
    i1 = i0 < in - 1 ? i0 + 1 : i0;
    l1 = s - (float)i0;
}

// dst[b][c][Y][X] = Σ taps · src[(b·h·w + y·w + x)·lds + c]
__global__ __launch_bounds__(256) void upsample_fwd_kernel(int B, int C, int h, int w, int H, int W, const float* __restrict__ src, long lds,
                                                           float* __restrict__ dst) {
    const size_t n = (size_t)B * C * H * W;
    const float sy = (float)h / H, sx = (float)w / W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int X = (int)(i % W), Y = (int)((i / W) % H), c = (int)((i / ((size_t)W * H)) % C), b = (int)(i / ((size_t)W * H * C));
        int y0, y1, x0, x1;
        float ly, lx;
        src_taps(Y, sy, h, y0, y1, ly);
        src_taps(X, sx, w, x0, x1, lx);
        const float* base = src + (size_t)b * h * w * lds + c;
        const float v00 = base[(size_t)(y0 * w + x0) * lds], v01 = base[(size_t)(y0 * w + x1) * lds];
        const float v10 = base[(size_t)(y1 * w + x0) * lds], v11 = base[(size_t)(y1 * w + x1) * lds];
        dst[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

// dsrc[(b·h·w + y·w + x)·lds + c] = Σ over the outputs that tap (y,x): a gather, so the sum order is fixed (no atomics).
// One thread per input pixel; separable weights: wy(Y) = contribution of input row y to output row Y.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(int B, int C, int h, int w, int H, int W, const float* __restrict__ dout,
                                                           float* __restrict__ dsrc, long lds) {
    const size_t n = (size_t)B * C * h * w;
    const float sy = (float)h / H, sx = (float)w / W;
    const int ry = (H + h - 1) / h + 1, rx = (W + w - 1) / w + 1;        // output rows/cols that can reach one input row/col, per side
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), c = (int)((i / ((size_t)w * h)) % C), b = (int)(i / ((size_t)w * h * C));
        const float* g = dout + ((size_t)b * C + c) * H * W;
        const int Yc = (int)((y + 0.5f) / sy), Xc = (int)((x + 0.5f) / sx);
        float acc = 0.f;
        for (int Y = max(0, Yc - 2 * ry); Y <= min(H - 1, Yc + 2 * ry); ++Y) {
            int y0, y1;
            float ly;
            src_taps(Y, sy, h, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int X = max(0, Xc - 2 * rx); X <= min(W - 1, Xc + 2 * rx); ++X) {
                int x0, x1;
                float lx;
                src_taps(X, sx, w, x0, x1, lx);
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx != 0.f) row = fmaf(wx, g[(size_t)Y * W + X], row);
            }
            acc = fmaf(wy, row, acc);
        }
        dsrc[((size_t)b * h * w + (size_t)y * w + x) * lds + c] = acc;
    }
}

// out[b][c] = scale · Σ_i x[(b·n + i)·ldx + c]      (forward: scale = 1/n)
__global__ __launch_bounds__(256) void segment_sum_kernel(int B, int n, int C, const float* __restrict__ x, long ldx, float scale, float* __restrict__ out) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f;
    for (int i = 0; i < n; ++i) a += x[((size_t)b * n + i) * ldx + c];
    out[(size_t)b * C + c] = a * scale;
}
// dx[(b·n + i)·ldx + c] = scale · dout[b][c]
__global__ __launch_bounds__(256) void segment_bcast_kernel(int B, int n, int C, const float* __restrict__ dout, float scale, float* __restrict__ dx, long ldx) {
    const size_t tot = (size_t)B * n * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t row = i / C;
        dx[row * ldx + c] = scale * dout[(row / n) * C + c];
    }
}

// ------------------------------------------------------------------------------------------ DiceCE
// MONAI DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr, smooth_dr) as the segmentation entry points use
// it (reference src/models/clipseg/segmentation.py:84, biomedclip/segmentation.py:75):
//   p = softmax_c(z);  dice[b,c] = 1 − (2·Σ p·t + nr) / (Σ p² + Σ t + dr);  loss = mean_{b,c} dice + mean_{b,pix}(−log p[label])
// Pass 1 (DICE_SLICES workgroups per image, fixed-order tree reduction each): I, P2, T per class and the CE sum → ws[b][slice][3·MAXC+1].
// Pass 2: dz_k = p_k·(g_k − Σ_c g_c p_c) + (p_k − t_k)/(B·HW),  g_c = (2/(B·C))·((2I_c+nr)·p_c/D_c² − t_c/D_c);  block 0 also
// reduces the loss.  C ≤ 8.
constexpr int DICE_MAXC = 8;

__device__ __forceinline__ void softmax_c(const float* __restrict__ z, size_t stride, int C, float (&p)[DICE_MAXC], float& lse) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < DICE_MAXC; ++c) if (c < C) { p[c] = z[c * stride]; m = fmaxf(m, p[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DICE_MAXC; ++c) if (c < C) { p[c] = __expf(p[c] - m); s += p[c]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < DICE_MAXC; ++c) if (c < C) p[c] *= inv;
    lse = m + __logf(s);
}

// DICE_SLICES workgroups per image (one per image left half of the chip idle and walked 50 176 pixels in 49 dependent iterations: 97 us for 128 images of 224 x 224);
// each writes its partial sums to ws[b][slice][3·MAXC + 1]; the second pass adds an image's slices up in a fixed order (deterministic, no atomics).
constexpr int DICE_SLICES = 16;
constexpr int DICE_NSUM = 3 * DICE_MAXC + 1;

__global__ __launch_bounds__(256) void dicece_sums_kernel(int C, int HW, const float* __restrict__ logits, const float* __restrict__ label,
                                                          float* __restrict__ ws) {
    __shared__ float red[4][DICE_NSUM];
    const int b = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* z = logits + (size_t)b * C * HW;
    const float* y = label + (size_t)b * HW;
    float acc[DICE_NSUM];
#pragma unroll
    for (int i = 0; i < DICE_NSUM; ++i) acc[i] = 0.f;
    for (int px = sl * 256 + tid; px < HW; px += DICE_SLICES * 256) {
        float p[DICE_MAXC], lse;
        softmax_c(z + px, (size_t)HW, C, p, lse);
        const int cls = (int)y[px];
#pragma unroll
        for (int c = 0; c < DICE_MAXC; ++c)
            if (c < C) {
                const float t = c == cls ? 1.f : 0.f;
                acc[c] += p[c] * t;
                acc[DICE_MAXC + c] += p[c] * p[c];
                acc[2 * DICE_MAXC + c] += t;
                if (c == cls) acc[3 * DICE_MAXC] += lse - z[(size_t)c * HW + px];
            }
    }
#pragma unroll
    for (int i = 0; i < DICE_NSUM; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < DICE_NSUM) ws[((size_t)b * DICE_SLICES + sl) * DICE_NSUM + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// an image's sums: its slices added in slice order
__device__ __forceinline__ float dice_sum(const float* __restrict__ ws, int b, int i) {
    const float* w = ws + (size_t)b * DICE_SLICES * DICE_NSUM + i;
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < DICE_SLICES; ++s) v += w[s * DICE_NSUM];
    return v;
}

__global__ __launch_bounds__(256) void dicece_grad_kernel(int B, int C, int HW, const float* __restrict__ logits, const float* __restrict__ label,
                                                          const float* __restrict__ ws, float nr, float dr, float* __restrict__ loss,
                                                          float* __restrict__ dlogits) {
    const int b = blockIdx.y;
    float a1[DICE_MAXC], a2[DICE_MAXC];                 // g_c = a1_c·p_c − a2_c·t_c
    const float k = 2.0f / ((float)B * C);
#pragma unroll
    for (int c = 0; c < DICE_MAXC; ++c)
        if (c < C) {
            const float D = dice_sum(ws, b, DICE_MAXC + c) + dice_sum(ws, b, 2 * DICE_MAXC + c) + dr;
            a1[c] = k * (2.f * dice_sum(ws, b, c) + nr) / (D * D);
            a2[c] = k / D;
        }
    const float ce_scale = 1.0f / ((float)B * HW);
    const float* z = logits + (size_t)b * C * HW;
    float* dz = dlogits + (size_t)b * C * HW;
    for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
        float p[DICE_MAXC], lse;
        softmax_c(z + px, (size_t)HW, C, p, lse);
        const int cls = (int)label[(size_t)b * HW + px];
        float gp = 0.f;
        float g[DICE_MAXC];
#pragma unroll
        for (int c = 0; c < DICE_MAXC; ++c)
            if (c < C) {
                g[c] = a1[c] * p[c] - (c == cls ? a2[c] : 0.f);
                gp = fmaf(g[c], p[c], gp);
            }
#pragma unroll
        for (int c = 0; c < DICE_MAXC; ++c)
            if (c < C) dz[(size_t)c * HW + px] = p[c] * (g[c] - gp) + ce_scale * (p[c] - (c == cls ? 1.f : 0.f));
    }
    if (b == 0 && blockIdx.x == 0) {                    // the loss: images dealt to the block's threads, then one fixed-order reduction
        __shared__ float lred[2][4];
        float dice = 0.f, ce = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) {
            for (int c = 0; c < C; ++c) dice += 1.0f - (2.f * dice_sum(ws, i, c) + nr) / (dice_sum(ws, i, DICE_MAXC + c) + dice_sum(ws, i, 2 * DICE_MAXC + c) + dr);
            ce += dice_sum(ws, i, 3 * DICE_MAXC);
        }
        dice = wave_sum(dice);
        ce = wave_sum(ce);
        if ((threadIdx.x & 63) == 0) { lred[0][threadIdx.x >> 6] = dice; lred[1][threadIdx.x >> 6] = ce; }
        __syncthreads();
        if (threadIdx.x == 0) {
            dice = (lred[0][0] + lred[0][1]) + (lred[0][2] + lred[0][3]);
            ce = (lred[1][0] + lred[1][1]) + (lred[1][2] + lred[1][3]);
            *loss = dice / ((float)B * C) + ce * ce_scale;
        }
    }
}

int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

int uia_upsample_bilinear_launch(hipStream_t stream, bool bwd, int B, int C, int h, int w, int H, int W, const float* in, float* out, long ld) {
    UIA_CHECK_ARG(B > 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0 && ld >= C, "uia_upsample_bilinear: bad shape B=%d C=%d %dx%d -> %dx%d ld=%ld", B, C, h, w, H, W, ld);
    UIA_CHECK_ARG(in && out, "uia_upsample_bilinear: null tensor");
    if (!bwd) hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, stream, B, C, h, w, H, W, in, ld, out);
    else hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((size_t)B * C * h * w)), dim3(256), 0, stream, B, C, h, w, H, W, in, out, ld);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_segment_mean_launch(hipStream_t stream, bool bwd, int B, int n, int C, const float* in, float* out, long ld) {
    UIA_CHECK_ARG(B > 0 && n > 0 && C > 0 && ld >= C, "uia_segment_mean: bad shape B=%d n=%d C=%d ld=%ld", B, n, C, ld);
    UIA_CHECK_ARG(in && out, "uia_segment_mean: null tensor");
    if (!bwd) hipLaunchKernelGGL(segment_sum_kernel, dim3((C + 255) / 256, B), dim3(256), 0, stream, B, n, C, in, ld, 1.0f / n, out);
    else hipLaunchKernelGGL(segment_bcast_kernel, dim3(grid_for((size_t)B * n * C)), dim3(256), 0, stream, B, n, C, in, 1.0f / n, out, ld);
    UIA_CHECK_LAUNCH();
    return 0;
}

size_t uia_dicece_ws_floats(int B) { return (size_t)B * DICE_SLICES * DICE_NSUM; }

int uia_dicece_launch(hipStream_t stream, int B, int C, int HW, const float* logits, const float* label, float nr, float dr, float* ws, float* loss,
                      float* dlogits) {
    UIA_CHECK_ARG(B > 0 && C >= 2 && C <= DICE_MAXC && HW > 0, "uia_dicece: bad shape B=%d C=%d HW=%d (2 <= C <= %d)", B, C, HW, DICE_MAXC);
    UIA_CHECK_ARG(logits && label && ws && loss && dlogits, "uia_dicece: null tensor");
    hipLaunchKernelGGL(dicece_sums_kernel, dim3(DICE_SLICES, B), dim3(256), 0, stream, C, HW, logits, label, ws);
    int gx = (HW + 255) / 256;
    gx = gx > 64 ? 64 : gx;
    hipLaunchKernelGGL(dicece_grad_kernel, dim3(gx, B), dim3(256), 0, stream, B, C, HW, logits, label, ws, nr, dr, loss, dlogits);
    UIA_CHECK_LAUNCH();
    return 0;
}
