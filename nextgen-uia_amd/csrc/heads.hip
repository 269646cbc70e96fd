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
