// elementwise.hip — the small HBM-bound layout kernels around the GEMMs.
//
//   cast / transpose_cast : fp32 master weights → T operand copies, [out,in] and its transpose
//                           (dgrad runs as a TN GEMM against Wᵀ, see gemm.hip)
//   im2col                : Conv2d(3, D, k=P, s=P) patch embedding as a GEMM operand
//                           (/root/reference/src/third_party/openai_clip/model.py:221,234; timm PatchEmbed)
//   fill_cls              : x[b,0,:] = class_embedding + pos[0]   (model.py:237-245; timm _pos_embed)
//   embed                 : x[b,l,:] = table[ids[b,l]] + pos[l] (+ type0)   (model.py:362-364; HF BertEmbeddings)
//   gather_rows           : out[i,:] = src[idx[i],:]            (EOT / CLS pooling, model.py:372)
//   scale                 : y = a*x (loss scaling of the incoming feature gradient)
#include "uia_common.h"
#include "uia_kernels.h"

namespace {

template <typename T>
__global__ void cast_kernel(size_t n4, const float* __restrict__ src, T* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        store4(dst + 4 * i, load4(src + 4 * i));
}

// dst[c][r] = src[r][c]; 32x32 tiles through LDS (padded), coalesced both ways
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(int rows, int cols, const float* __restrict__ src, T* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * rows + r] = from_f32<T>(tile[tx][j]);
    }
}

// every operand form of matrix blockIdx.y (uia_pack_desc) from one read of its fp32 source; the matrices are adapter-sized
// (≤ a few hundred KB), so the scattered 2-byte stores of the transposed / K-blocked forms stay inside L2
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const uia_pack_desc* __restrict__ descs) {
    const uia_pack_desc d = descs[blockIdx.y];
    constexpr int g = 64 / (int)sizeof(T);
    const int R = d.rows, C = d.cols, total = R * C;
    const int RP = d.rows_pad > 0 ? d.rows_pad : R, CP = d.cols_pad > 0 ? d.cols_pad : C;      // destination extents (zero padding is the caller's)
    const float sc = d.scale != 0.f ? d.scale : 1.f;
    T* row = (T*)d.row; T* row_kb = (T*)d.row_kb; T* tr = (T*)d.tr; T* tr_kb = (T*)d.tr_kb;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int r = e / C, c = e - r * C;
        const T v = from_f32<T>(d.src[e] * sc);
        if (row) row[(size_t)r * CP + c] = v;
        if (row_kb) row_kb[((size_t)(c / g) * RP + r) * g + (c % g)] = v;
        if (tr) tr[(size_t)c * RP + r] = v;
        if (tr_kb) tr_kb[((size_t)(r / g) * CP + c) * g + (r % g)] = v;
    }
}

// out[(b*gh+py)*gw+px][(c*P+ky)*P+kx] = img[b][c][py*P+ky][px*P+kx]; one thread = 4 consecutive kx
template <typename T>
__global__ void im2col_kernel(int B, int C, int H, int W, int P, const float* __restrict__ img, T* __restrict__ out) {
    const int gh = H / P, gw = W / P, K = C * P * P, K4 = K >> 2;
    const size_t total = (size_t)B * gh * gw * K4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const size_t prow = i / K4;
        const int px = (int)(prow % gw), py = (int)((prow / gw) % gh), b = (int)(prow / ((size_t)gw * gh));
        const int k = k4 * 4, kx = k % P, ky = (k / P) % P, c = k / (P * P);
        const float* s = img + (((size_t)b * C + c) * H + (size_t)py * P + ky) * W + (size_t)px * P + kx;
        store4(out + prow * K + k, load4(s));
    }
}

// any patch size, output rows padded to `ldo` columns (zeros beyond C·P·P): ViT-L/14's 588-wide patches → 640 for the GEMM's K granule
template <typename T>
__global__ void im2col_padded_kernel(int B, int C, int H, int W, int P, const float* __restrict__ img, T* __restrict__ out, long ldo) {
    const int gh = H / P, gw = W / P, K = C * P * P;
    const size_t total = (size_t)B * gh * gw * ldo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % ldo);
        const size_t prow = i / ldo;
        float v = 0.f;
        if (k < K) {
            const int px = (int)(prow % gw), py = (int)((prow / gw) % gh), b = (int)(prow / ((size_t)gw * gh));
            const int kx = k % P, ky = (k / P) % P, c = k / (P * P);
            v = img[(((size_t)b * C + c) * H + (size_t)py * P + ky) * W + (size_t)px * P + kx];
        }
        out[i] = from_f32<T>(v);
    }
}

__global__ void fill_cls_kernel(int B, int N, int D, const float* __restrict__ cls, const float* __restrict__ pos0, float* __restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    x[(size_t)b * N * D + d] = cls[d] + (pos0 ? pos0[d] : 0.f);
}

// An id outside [0, vocab) (nn.Embedding raises for it) never reads the table: its row is filled with NaN, which reaches the
// loss and trips the loops' non-finite check instead of silently training on whatever lies beyond the table.
__global__ void embed_kernel(int rows, int L, int D, int vocab, const int64_t* __restrict__ ids, const float* __restrict__ table,
                             const float* __restrict__ pos, const float* __restrict__ type0, float* __restrict__ out) {
    const int D4 = D >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * D4; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / D4), c = (int)(i % D4) * 4, l = r % L;
        const int64_t id = ids[r];
        if (id < 0 || id >= vocab) {
            const float q = __builtin_nanf("");
            store4(out + (size_t)r * D + c, f32x4{q, q, q, q});
            continue;
        }
        f32x4 v = load4(table + (size_t)id * D + c);
        const f32x4 pv = load4(pos + (size_t)l * D + c);
        v += pv;
        if (type0) v += load4(type0 + c);
        store4(out + (size_t)r * D + c, v);
    }
}

// packed rows: out[r] = table[ids[r]] + pos[pos_idx[r]] (+ type0)   (un-padded text tower: r runs over the valid tokens only)
__global__ void embed_packed_kernel(int rows, int D, int vocab, int max_pos, const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_idx,
                                    const float* __restrict__ table, const float* __restrict__ pos, const float* __restrict__ type0,
                                    float* __restrict__ out) {
    const int D4 = D >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * D4; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / D4), c = (int)(i % D4) * 4;
        const int64_t id = ids[r], pi = pos_idx[r];
        if (id < 0 || id >= vocab || pi < 0 || pi >= max_pos) {
            const float q = __builtin_nanf("");
            store4(out + (size_t)r * D + c, f32x4{q, q, q, q});
            continue;
        }
        f32x4 v = load4(table + (size_t)id * D + c);
        v += load4(pos + (size_t)pi * D + c);
        if (type0) v += load4(type0 + c);
        store4(out + (size_t)r * D + c, v);
    }
}

// d table[ids[r]] += dx[r]   (nn.Embedding backward; rows whose id is `pad_id` contribute nothing: padding_idx semantics)
__global__ void embed_bwd_kernel(int rows, int D, int vocab, const int64_t* __restrict__ ids, const float* __restrict__ dx, float* __restrict__ dtable, long pad_id) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * D; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / D), c = (int)(i % D);
        const int64_t id = ids[r];
        if (id != pad_id && id >= 0 && id < vocab) atomicAdd(dtable + (size_t)id * D + c, dx[i]);
    }
}

__global__ void gather_rows_kernel(int n, int D, const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst) {
    const int D4 = D >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * D4; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / D4), c = (int)(i % D4) * 4;
        store4(dst + (size_t)r * D + c, load4(src + (size_t)idx[r] * D + c));
    }
}

template <typename T>
__global__ void scale_cast_kernel(size_t n4, float a, const float* __restrict__ src, T* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = load4(src + 4 * i);
        v *= a;
        store4(dst + 4 * i, v);
    }
}

// dst = (accumulate ? dst : 0) + src * keep(seed, idx) / (1 - p)      (LoRA input dropout, lora.py:82-83, and its backward)
// One draw (dropout_keep8) per eight elements: the same generator the fused forms use (N = 64 stream kernel, run-time GEMM epilogue).
template <typename T>
__global__ void dropout_kernel(size_t n8, const T* __restrict__ src, T* __restrict__ dst, float inv_keep, uint32_t thresh16, uint64_t seed, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float v[8], o[8];
        load8(src + 8 * i, v);
        if (accumulate) load8(dst + 8 * i, o);
        const uint32_t keep = dropout_keep8(seed, (uint32_t)i, thresh16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float k = (keep >> e) & 1u ? inv_keep : 0.f;
            o[e] = (accumulate ? o[e] : 0.f) + v[e] * k;
        }
        store8(dst + 8 * i, o);
    }
}

// out[n] += Σ_m A[m][n]: block = 64 columns x 4 row-quarters of a 1024-row chunk
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const T* __restrict__ A, long lda, float* __restrict__ out) {
    __shared__ float red[256];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 1024, m1 = min(M, m0 + 1024);
    float s = 0.f;
    if (col < N)
        for (int m = m0 + part; m < m1; m += 4) s += to_f32(A[(size_t)m * lda + col]);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && col < N) atomicAdd(out + col, red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x]);
}

inline int grid_for(size_t work, int block) {
    size_t g = (work + block - 1) / block;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

int uia_cast_launch(hipStream_t stream, int dtype, size_t n, const float* src, void* dst, float scale) {
    UIA_CHECK_ARG(n % 4 == 0 && src && dst, "uia_cast: n=%zu must be a multiple of 4 and tensors non-null", n);
    UIA_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "uia_cast: alignment");
    const int g = grid_for(n / 4, 256);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(scale_cast_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, n / 4, scale, src, (bf16_t*)dst);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(scale_cast_kernel<float>, dim3(g), dim3(256), 0, stream, n / 4, scale, src, (float*)dst);
    else { uia_set_error("uia_cast: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_transpose_cast_launch(hipStream_t stream, int dtype, int rows, int cols, const float* src, void* dst) {
    UIA_CHECK_ARG(rows > 0 && cols > 0 && src && dst, "uia_transpose_cast: bad arguments");
    const dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, dim3(256), 0, stream, rows, cols, src, (bf16_t*)dst);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, stream, rows, cols, src, (float*)dst);
    else { uia_set_error("uia_transpose_cast: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_pack_weights_launch(hipStream_t stream, int dtype, int n, const uia_pack_desc* descs_device, int max_elems) {
    UIA_CHECK_ARG(n > 0 && n <= 65535 && descs_device && max_elems > 0, "uia_pack_weights: bad arguments (n=%d, max_elems=%d)", n, max_elems);
    int bx = (max_elems + 1023) / 1024;                     // ~4 elements per thread
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(bx, n), dim3(256), 0, stream, descs_device);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(bx, n), dim3(256), 0, stream, descs_device);
    else { uia_set_error("uia_pack_weights: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_im2col_launch(hipStream_t stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out) {
    UIA_CHECK_ARG(B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && P % 4 == 0, "uia_im2col: unsupported geometry B=%d C=%d H=%d W=%d P=%d", B, C, H, W, P);
    UIA_CHECK_ARG(img && out && (uintptr_t)img % 16 == 0 && W % 4 == 0, "uia_im2col: null or misaligned tensor");
    const size_t work = (size_t)B * (H / P) * (W / P) * (C * P * P / 4);
    const int g = grid_for(work, 256);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, B, C, H, W, P, img, (bf16_t*)out);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(g), dim3(256), 0, stream, B, C, H, W, P, img, (float*)out);
    else { uia_set_error("uia_im2col: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_im2col_padded_launch(hipStream_t stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out, long ldo) {
    UIA_CHECK_ARG(B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && ldo >= (long)C * P * P, "uia_im2col_padded: unsupported geometry B=%d C=%d H=%d W=%d P=%d ldo=%ld", B, C, H, W, P, ldo);
    UIA_CHECK_ARG(img && out, "uia_im2col_padded: null tensor");
    const int g = grid_for((size_t)B * (H / P) * (W / P) * ldo, 256);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(im2col_padded_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, B, C, H, W, P, img, (bf16_t*)out, ldo);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(im2col_padded_kernel<float>, dim3(g), dim3(256), 0, stream, B, C, H, W, P, img, (float*)out, ldo);
    else { uia_set_error("uia_im2col_padded: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_fill_cls_launch(hipStream_t stream, int B, int N, int D, const float* cls, const float* pos0, float* x) {
    UIA_CHECK_ARG(B > 0 && N > 0 && D > 0 && cls && x, "uia_fill_cls: bad arguments");
    hipLaunchKernelGGL(fill_cls_kernel, dim3((B * D + 255) / 256), dim3(256), 0, stream, B, N, D, cls, pos0, x);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_embed_launch(hipStream_t stream, int rows, int L, int D, int vocab, int max_pos, const int64_t* ids, const float* table, const float* pos, const float* type0, float* out) {
    UIA_CHECK_ARG(rows > 0 && L > 0 && D % 4 == 0 && vocab > 0 && ids && table && pos && out, "uia_embed: bad arguments");
    UIA_CHECK_ARG(L <= max_pos, "uia_embed: sequence length %d exceeds the position table (%d rows)", L, max_pos);
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for((size_t)rows * D / 4, 256)), dim3(256), 0, stream, rows, L, D, vocab, ids, table, pos, type0, out);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_embed_packed_launch(hipStream_t stream, int rows, int D, int vocab, int max_pos, const int64_t* ids, const int64_t* pos_idx, const float* table, const float* pos,
                             const float* type0, float* out) {
    UIA_CHECK_ARG(rows > 0 && D % 4 == 0 && vocab > 0 && max_pos > 0 && ids && pos_idx && table && pos && out, "uia_embed_packed: bad arguments");
    hipLaunchKernelGGL(embed_packed_kernel, dim3(grid_for((size_t)rows * D / 4, 256)), dim3(256), 0, stream, rows, D, vocab, max_pos, ids, pos_idx, table, pos, type0, out);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_embed_bwd_launch(hipStream_t stream, int rows, int D, int vocab, const int64_t* ids, const float* dx, float* dtable, long pad_id) {
    UIA_CHECK_ARG(rows > 0 && D > 0 && vocab > 0 && ids && dx && dtable, "uia_embed_bwd: bad arguments");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((size_t)rows * D, 256)), dim3(256), 0, stream, rows, D, vocab, ids, dx, dtable, pad_id);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_gather_rows_launch(hipStream_t stream, int n, int D, const float* src, const int64_t* idx, float* dst) {
    UIA_CHECK_ARG(n > 0 && D % 4 == 0 && src && idx && dst, "uia_gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)n * D / 4, 256)), dim3(256), 0, stream, n, D, src, idx, dst);
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_dropout_launch(hipStream_t stream, int dtype, size_t n, const void* src, void* dst, float p, uint64_t seed, int accumulate) {
    UIA_CHECK_ARG(n % 8 == 0 && src && dst && p >= 0.f && p < 1.f, "uia_dropout: bad arguments (n=%zu, p=%f)", n, p);
    const float inv_keep = 1.0f / (1.0f - p);
    UIA_CHECK_ARG(n / 8 <= 0xFFFFFFFFull, "uia_dropout: n=%zu exceeds the generator's 2^35 elements", n);
    const uint32_t thresh = dropout_thresh16(p);
    const int g = grid_for(n / 8, 256);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, n / 8, (const bf16_t*)src, (bf16_t*)dst, inv_keep, thresh, seed, accumulate);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(g), dim3(256), 0, stream, n / 8, (const float*)src, (float*)dst, inv_keep, thresh, seed, accumulate);
    else { uia_set_error("uia_dropout: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}

int uia_colsum_launch(hipStream_t stream, int dtype, int M, int N, const void* A, long lda, float* out) {
    UIA_CHECK_ARG(M > 0 && N > 0 && A && out && lda >= N, "uia_colsum: bad arguments");
    const dim3 grid((N + 63) / 64, (M + 1023) / 1024);
    if (dtype == UIA_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, stream, M, N, (const bf16_t*)A, lda, out);
    else if (dtype == UIA_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, M, N, (const float*)A, lda, out);
    else { uia_set_error("uia_colsum: bad dtype %d", dtype); return -1; }
    UIA_CHECK_LAUNCH();
    return 0;
}
