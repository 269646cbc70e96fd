/* uia_hip.h — C ABI of libuia_hip.so: the MI355X (gfx950) kernels behind the CLIP-adapter
 * fine-tune hot path of jinggqu/NextGen-UIA.
 *
 * The reference is pure Python on stock PyTorch and has NO native boundary of its own; what a
 * maintainer would bind is the set of ATen calls its hot path makes.  Each entry point below names
 * the reference call sites (file:line under /root/reference) whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding and the module-level swap-in.
 *
 * Contract (every function):
 *   - extern "C", returns 0 on success, <0 on error; uia_last_error() gives a thread-local message.
 *   - no allocation inside: the caller owns every buffer (PyTorch device tensors → data_ptr()).
 *   - asynchronous on the hipStream_t passed as `stream` (torch.cuda.current_stream().cuda_stream);
 *     no internal threads, no host synchronisation, graph-capturable.
 *   - one process per GPU; the only global state is the RCCL communicator of uia_comm_*.
 *   - dtype: UIA_F32 (parity mode, exact-fp32 MFMA / VALU) or UIA_BF16 (operands bf16, fp32
 *     accumulate).  "T" below means that element type.  The residual stream, losses, parameter
 *     gradients and optimiser state are fp32 in both modes.
 *   - tensors are row-major and contiguous unless a leading dimension (in elements) is given;
 *     16-byte alignment of every base pointer and row is required and checked.
 */
#ifndef UIA_HIP_H
#define UIA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { UIA_F32 = 0, UIA_BF16 = 1 };
enum { UIA_ACT_NONE = 0, UIA_ACT_GELU = 1, UIA_ACT_QUICKGELU = 2, UIA_ACT_RELU = 3 };
enum { UIA_MASK_NONE = 0, UIA_MASK_CAUSAL = 1, UIA_MASK_KEYPAD = 2 };
enum { UIA_MONA_BASELINE = 0, UIA_MONA_NOISE_AWARE = 1, UIA_MONA_FREQ_ENHANCED = 2, UIA_MONA_HYBRID = 3 };

const char* uia_last_error(void);
int uia_version(void);

/* ---------------------------------------------------------------------------------------------
 * Dense contraction  C[M,N] = epilogue(alpha · A[M,K] · W[N,K]ᵀ)   (both operands K-contiguous).
 * Replaces every nn.Linear / F.linear / addmm of the path:
 *   src/third_party/openai_clip/model.py:181-188,197,200-201,255,372  (in_proj, out_proj, c_fc, c_proj, proj)
 *   src/adapters/mona.py:127,148 (project1 / project2)      src/adapters/lora.py:80,87 (base + rank factors)
 *   timm Attention.qkv/proj, Mlp.fc1/fc2; HF BertSelfAttention / BertIntermediate / BertOutput  [third-party]
 * and, run against the cached transpose Wᵀ, their autograd dgrad  dx = dy·W.
 * Epilogue order: v = alpha·acc + bias;  aux_out ← v;  v = act(v);  v *= act'(aux_in)  (dact);
 *                 v += resid (fp32);  v += residT (T);  out32 ← v;  outT ← v.
 * Requirements: K % (128/sizeof(T)) == 0, N % 8 == 0. */
typedef struct uia_gemm_desc {
    const void* A; int64_t lda;
    const void* W; int64_t ldw;
    int32_t M, N, K;
    float alpha;
    const float* bias;
    int32_t act;
    int32_t dact;
    const void* aux_in; int64_t ldaux_in;
    void* aux_out; int64_t ldaux_out;
    const float* resid; int64_t ldr;
    int32_t resid_mod, resid_row_off;   /* resid_mod>0: residual row = m % resid_mod + resid_row_off (pos-embed) */
    const void* residT; int64_t ldrT;
    int32_t out_group;                  /* >0: output row = m + m/out_group + 1 (patch rows -> token rows, CLS slot skipped) */
    void* outT; int64_t ldo;
    float* out32; int64_t ldo32;
    int32_t w_kblocked;                 /* 1: W is stored K-blocked, [K/g][N][g] with g = 64 bytes / sizeof(T) elements (ldw ignored): the layout
                                           the ring tile configs (8, 10; the automatic choice for M > 2048, N > 64) stream fastest; other tile
                                           configs reject it */
    const void* resid_ln_stats;         /* non-null: `resid` holds the INPUT of a LayerNorm and the residual added is that LayerNorm's output,
                                           fmaf((resid[m][n] - mean_m)·rstd_m, resid_ln_w[n], resid_ln_b[n]) with (mean_m, rstd_m) =
                                           resid_ln_stats[2m], [2m+1] as written by uia_layernorm_fwd_stats: a post-LN (BERT) sub-layer sum
                                           then reads the previous sum once instead of the LayerNorm writing its fp32 output for it
                                           (HF BertSelfOutput / BertOutput: LayerNorm(dense(h) + input_tensor) [third-party]) */
    const float* resid_ln_w;
    const float* resid_ln_b;
    /* LayerNorm folded into the GEMMs on either side of it (frozen affine; timm Block norm1 / norm2 and HF BertSelfOutput / BertOutput
     * LayerNorm [third-party], model.py:163-169 + 181-202 for the OpenAI blocks): the PRODUCER of the normalised rows x adds their row
     * sums to `rowsum_out` while it stores them (and can write their T copy through outT beside out32); the CONSUMER takes that T copy of
     * the RAW rows as A and a weight whose columns are pre-scaled by the LayerNorm weight, W'[n][k] = W[n][k]·ln_w[k], and evaluates
     *     LN(x)·Wᵀ + b  =  rstd_m·(x·W'ᵀ − mean_m·colsum[n]) + (b + W·ln_b)[n]
     * in its epilogue (the caller passes colsum[n] = Σ_k W'[n][k] and the combined bias as `bias`), so the stand-alone LayerNorm pass
     * over the rows (read 4 B, write 2 B per element) disappears. */
    int64_t* rowsum_out;                /* non-null: rowsum_out[2m] += Σ_n v[m][n], rowsum_out[2m+1] += Σ_n v[m][n]² over the N columns of the fp32
                                           result v this launch stores, as 64-bit FIXED-POINT numbers in units of 2^-30 (integer atomics: the
                                           sums do not depend on the order in which a row's column tiles arrive; caller zeroes; 16-byte aligned) */
    const int64_t* lnfold_sums;         /* non-null: A holds raw rows; (Σ, Σ²) of row m over lnfold_dim columns at lnfold_sums[2m], [2m+1],
                                           in the fixed-point form rowsum_out leaves them */
    const float* lnfold_colsum;         /* [N] fp32 column sums of the pre-scaled weight */
    int32_t lnfold_dim; float lnfold_eps;
    int32_t resid_ln_dim; float resid_ln_eps;   /* resid_ln_dim > 0: resid_ln_stats points at (Σ, Σ²) over resid_ln_dim columns as rowsum_out leaves
                                                   them (int64 fixed point), not at float (mean, rstd); mean = Σ/dim, rstd = rsqrt(max(Σ²/dim − mean², 0) + resid_ln_eps) */
    /* K-BLOCKED ACTIVATIONS (ring tile configs with 64-byte sub-tiles: 8, 10, 13, 14; the other configs reject them).  The operand of a
     * large GEMM streams fastest as [K·sizeof(T)/64][rows][64 bytes] (each 1 KiB LDS-DMA piece = 8 whole 128-byte lines; +1.5…3.4 % per
     * launch on the step's shapes, profiles/r02_a_gemm_order_layout.txt).  A GEMM whose T result is the next GEMM's A operand can write it
     * in that layout directly, and the consumer reads it with a_kb_rows set:
     *   a_kb_rows    > 0: A is K-blocked; element (m, k) is at A[((k / g)·a_kb_rows + m)·g + k % g], g = 64 / sizeof(T) (lda ignored).
     *                     a_kb_rows is the row count of the WHOLE K-blocked tensor (its plane stride), A may point at a row offset in it.
     *   outT_kb_rows > 0: the T output is written K-blocked the same way, column n playing the part of k (N·sizeof(T) % 64 == 0; ldo ignored). */
    int64_t a_kb_rows;
    int64_t outT_kb_rows;
    /* Guard of the folded LayerNorms (optional; a caller-owned, caller-zeroed device word that many launches may share).  The fold hands
     * the GEMM bf16(x) instead of bf16(LN(x)): fine for roughly centred rows, lossy for |mean| >> std.  Every launch that carries
     * rowsum_out / lnfold_sums / resid_ln_dim checks the rows it touches and ORs into *ln_flag
     *   bit 0: a consumer saw a row with |mean|·rstd > ln_flag_limit   (the caller should go back to the stand-alone LayerNorm kernels)
     *   bit 1: a producer's partial row sum was non-finite or outside the fixed-point range (|Σ| or Σ² >= 5e8 per wave column; the partial
     *          is clamped so that the integer atomics cannot wrap, and the statistics of that row are garbage: the caller must fail loudly)
     * The atomics are issued only for offending rows, so a healthy step pays one compare per row. */
    int32_t* ln_flag;
    float ln_flag_limit;                /* <= 0: 8.0 */
    /* LoRA input dropout (/root/reference/src/adapters/lora.py:82-83) without a pass of its own.  keep(seed, element index) is the generator
     * of uia_dropout (eight consecutive elements per draw), so a fused launch and uia_dropout + a plain launch see the same mask.
     *   drop_where = 1: the A operand is dropped while it streams through the N = 64 kernel (tile cfg 16; element (m, k) has index m·K + k);
     *                   a_drop_out, if set, receives the dropped rows ([M, K], leading dimension lda) for the weight gradient.
     *   drop_where = 2: alpha·acc (+ bias, activation) of output element (m, n), index m·N + n, is dropped before the residual adds:
     *                   dx += drop(s·q·A), the backward of the same dropout (run-time epilogue; N % 8 == 0). */
    int32_t drop_where;
    float drop_p;
    uint64_t drop_seed;
    void* a_drop_out;
    /* SPLIT K for launches of a few tiles and a long K (the M tail of a large GEMM: its cost is the latency of the K chain, not work).  Two calls with
     * the SAME descriptor on tile config 13 (half-height tiles, run-time epilogue): 13 | slices << 16 | 1 << 22 runs one workgroup per (tile, K slice),
     * each adding its raw fp32 accumulators into the tile's image in splitk_ws (hardware float atomics); 13 | slices << 16 | 2 << 22 reads the image,
     * zeroes it again and runs the epilogue.  splitk_ws: tiles · 128·256 floats (tiles = ceil(M/128)·ceil(N/256)), 16-byte aligned, caller-owned,
     * ZERO before the first use (the launches keep it zero between uses).  The slice partials meet in no fixed order: last-bit run-to-run variation. */
    float* splitk_ws;
    /* K EXTENSION: the LoRA rank update inside the frozen GEMM (y = x·Wᵀ + s·t·Bᵀ as ONE K loop over [x | t]·[W | s·B]ᵀ; lora.py:87).  K counts BOTH parts
     * and W holds K columns; A supplies the first K − K2 of them (row-major, lda), A2 the last K2: a row-major [M, K2] operand (lda2) per group of
     * a2_group_cols output columns, group g at A2 + g·a2_group_stride elements (a fused q | k | v projection has three; 0 = one operand for all columns).
     * bf16, tile cfgs 8 / 13, K2 % 32 == 0, a2_group_cols % 256 == 0. */
    const void* A2;
    int64_t lda2;
    int32_t K2;
    int32_t a2_group_cols;
    int64_t a2_group_stride;
    /* THREE-BYTE fp32 tensors for the HBM-bound epilogues of a frozen post-LN tower (bf16 launches): a value x is kept as hi = bf16(x) (round to
     * nearest: the T copy the next GEMM reads as its A operand anyway) plus lo = the next 8 mantissa bits as a signed byte,
     *     float_bits(x) ≈ (hi_bits << 16) + (lo << 8)      (15 stored mantissa bits: 2^-16 relative),
     * so a sub-layer sum costs 3 + 3 bytes per element in the epilogue that reads one and writes the next, instead of 4 + (4 + 2).
     *   resid_lo8 != NULL: the residual is (residT, resid_lo8) in that form — residT row-major (ldrT) or, residT_kb_rows > 0, K-blocked like
     *                      outT; `resid` must be NULL; resid_ln_* (stats / weight / bias / dim) then apply to the reconstructed rows.
     *   out_lo8   != NULL: the result's low bytes go there (row-major, ld_out_lo), its hi plane is outT (required); out32 may be NULL.
     *   resid_lo_kb_rows / out_lo_kb_rows > 0: that plane of low bytes is column-blocked like the K-blocked hi plane, 64 columns (64 bytes) per
     *                      block with that many rows: byte (m, n) at ((n / 64)·rows + m)·64 + n % 64 — a 16-row pass of the epilogue then
     *                      moves whole 128-byte lines instead of 64-byte pieces 768 bytes apart (N % 64 == 0). */
    const int8_t* resid_lo8;
    int64_t ld_resid_lo;
    int64_t residT_kb_rows;
    int8_t* out_lo8;
    int64_t ld_out_lo;
    int64_t resid_lo_kb_rows;
    int64_t out_lo_kb_rows;
} uia_gemm_desc;
/* tile_cfg: 0 = chosen from the shape (what every caller in this repo passes unless it runs an experiment).  Low byte = a kernel: 8 the 256 x 256 ring tile (eight
 * waves, LDS-DMA), 13 / 14 its half-height forms, 12 persistent, 16 / 23 the N = 64 / K = 64 streams, 24 five-deep ring, 25 / 26 four waves of 128 x 128, 27 / 28 / 29 the same
 * with operands staged through registers (two / three sub-tiles in flight / a persistent grid; bf16, no K extension, operands below 4 GiB), 1-5 / 21 small tiles; bits 8-15 =
 * row panels per tile-order group (255 = none).  Every ring kernel returns bit-identical results for a given descriptor; an unsupported combination returns -1. */
int uia_gemm(void* stream, int dtype, const uia_gemm_desc* d, int tile_cfg /* 0 = auto */);

/* Parameter gradient of a Linear:  dW[I,J] += Σ_m A[m,I]ᵀ·B[m,J]  and optionally dbias[I] += Σ_m A[m,I]
 * (fp32 accumulate with atomics into caller-zeroed buffers).  One of I, J must be ≤ 64·k.
 * Replaces autograd's wgrad of mona.py:127,148 and lora.py:87. */
int uia_wgrad(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb,
              float alpha, float* dW, float* dbias_A);
/* The same with the operands zero-padded to the kernel's 64-wide tiles and the gradient not: only rows < i_valid and columns < j_valid are
 * accumulated, into dW with leading dimension ldw — a LoRA factor's [out, r] / [r, in] gradient (r = 16 padded to 64) lands in the
 * parameter's own .grad without a padded staging buffer, its zero fill and its slice copy (lora.py:87). */
int uia_wgrad_ex(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb,
                 float alpha, float* dW, int64_t ldw, int i_valid, int j_valid, float* dbias_A);
/* dW[I,J] += alpha * A.T @ drop(B): the LoRA factor gradient dA = s * q.T @ dropout(x) (lora.py:82-87) with the dropout REGENERATED while B is staged —
 * B holds the un-dropped rows, a window of J columns starting at column drop_col0 of a [M, drop_ld] tensor whose mask uia_dropout /
 * uia_gemm_desc.drop_where = 1 draw from (seed, element index / 8); the forward then need not write the dropped rows out.  bf16; extents as uia_wgrad_ex. */
int uia_wgrad_drop(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb,
                   float alpha, float* dW, int64_t ldw, int i_valid, int j_valid, float drop_p, uint64_t seed, int64_t drop_ld, int drop_col0);
/* Up to four weight gradients of ONE shape in one launch (the q | k | v factors of a LoRA attention block, lora.py:82-87 three times: three launches of short-lived workgroups
 * were three latencies).  Problem g: dW[g][i_valid, j_valid] += alpha * A[g].T @ drop_g(B[g]) (+ dbias_A[g] += column sums of A[g] where given, without dropout only); all problems
 * share M, I, J, the leading dimensions, alpha, the valid extent and the dropout window; drop_p = 0: no dropout, else problem g's mask is drawn from drop_seed[g] as uia_wgrad_drop
 * draws it.  bf16.  Same arithmetic per problem as uia_wgrad_ex / uia_wgrad_drop. */
typedef struct uia_wgrad_group_desc {
    int32_t n, M, I, J;
    const void* A[4];
    const void* B[4];
    float* dW[4];
    float* dbias_A[4];
    int64_t lda, ldb, ldw;
    int32_t i_valid, j_valid;
    float alpha, drop_p;
    uint64_t drop_seed[4];
    int64_t drop_ld;
    int32_t drop_col0, reserved;
} uia_wgrad_group_desc;
int uia_wgrad_group(void* stream, int dtype, const uia_wgrad_group_desc* d);

/* ---------------------------------------------------------------------------------------------
 * softmax(q kᵀ·scale + mask) v, head dim 64, L <= 272, one workgroup per (batch, head).
 * Replaces nn.MultiheadAttention (model.py:195-197), F.scaled_dot_product_attention
 * (src/adapters/lora.py:188; timm Attention [third-party]), HF BertSelfAttention [third-party].
 * Element (b,l,h,d) of q/k/v is ptr[(b*L+l)*ld_qkv + h*64 + d]; likewise out/dout/dq/dk/dv. */
typedef struct uia_attn_desc {
    const void *q, *k, *v; int64_t ld_qkv;
    void* out; int64_t ldo;
    float* lse;               /* [B,H,L] fp32 log-sum-exp of the scaled scores (fwd: written if non-null) */
    const int32_t* keylen;    /* [B] valid keys (UIA_MASK_KEYPAD) */
    int32_t B, H, L, dh;
    int32_t mask_kind;
    float scale;
    const void* dout; int64_t lddo;              /* backward only */
    void *dq, *dk, *dv; int64_t ld_dqkv;         /* backward only */
    const int32_t* cu_seqlens; /* forward only, optional: [B+1] row offsets of PACKED (un-padded) sequences; sequence b then has
                                  cu[b+1]-cu[b] <= L tokens at rows cu[b].. and every key is valid (L is the maximum length) */
    /* bf16, dh = 64: K-blocked tensors on the GEMM side of the attention (uia_gemm_desc.a_kb_rows; g = 32 elements).
     *   out_kb_rows  > 0: `out` is K-blocked with that many rows — written so by the forward (it is the A operand of the output
     *                     projection), read so by the backward; element (row, h, d) is at out[((2h + d/32)·out_kb_rows + row)·32 + d%32].
     *   dqkv_kb_rows > 0: backward: dq, dk, dv are written K-blocked the same way (dq / dk / dv each point at the first column block
     *                     of their part of the fused [rows, 3·H·64] gradient, the A operand of the QKV data-gradient GEMM). */
    int64_t out_kb_rows;
    int64_t dqkv_kb_rows;
} uia_attn_desc;
int uia_attn_fwd(void* stream, int dtype, const uia_attn_desc* d);
int uia_attn_bwd(void* stream, int dtype, const uia_attn_desc* d);
/* Same backward with an explicit kernel configuration (bf16, dh = 64; ignored otherwise): 0 = the library's choice (what uia_attn_bwd
 * runs), 1 = the lock-step 8-wave kernel (dS crosses LDS once per 32-query block), 2 / 3 / 4 = the barrier-free unit kernel (waves pull
 * key-tile and query-tile units from an LDS counter; no product of one wave is read by another) with 8 waves and V in LDS / 4 waves and
 * V fragments from global memory (two heads per CU up to 208 tokens) / 8 waves and V from global memory (2 stages the O rows that δ needs
 * through LDS up to 240 tokens; 6 = 2 with those operands from global memory at every length); 5 = the persistent form of
 * the unit kernel (one workgroup per CU walks the heads, the next head's operands land under the current head's sweeps; L <= 240).
 * For A/B timing and tests. */
int uia_attn_bwd_cfg(void* stream, int dtype, const uia_attn_desc* d, int cfg);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over fp32 rows (model.py:163-169; timm / HF LayerNorm [third-party]).
 * x rows may be strided by ldx (elements); y / dy are compact [M,D].  Backward is for FROZEN
 * gamma/beta (dx only): dx = dres + LN'(dy); statistics are recomputed from x. */
int uia_layernorm_fwd(void* stream, int dtype, int M, int D, int64_t ldx, const float* x, const float* gamma, const float* beta,
                      float eps, void* yT, float* y32);
/* Same, and additionally stats[2m] = mean_m, stats[2m+1] = rstd_m (fp32) when `stats` is non-null; yT and y32 may then both be null
 * only if stats is given.  The deferred-residual form of uia_gemm (resid_ln_stats) consumes them. */
int uia_layernorm_fwd_stats(void* stream, int dtype, int M, int D, int64_t ldx, const float* x, const float* gamma, const float* beta,
                            float eps, void* yT, float* y32, float* stats);
int uia_layernorm_bwd(void* stream, int dtype, int M, int D, int64_t ldx, const void* dy, const float* x, const float* gamma,
                      float eps, const float* dres, float* dx32, void* dxT);
/* The same backward on THREE-BYTE tensors (bf16 launches, compact rows; the form is defined at uia_gemm_desc.resid_lo8): inside a frozen block
 * the attention-half output x1 and its gradient dx1 never exist in fp32 — x1 leaves the output projection's epilogue as (T copy, low bytes) and
 * is read here for the statistics; dx1 leaves this kernel as (dxT, dx_lo) and comes back as the next call's (dres_hi, dres_lo).
 *   x_lo   != NULL: x is (x_hi, x_lo); x must be NULL; x_hi row-major [M, D] or, x_kb_rows > 0, K-blocked with that many rows per 32-column block.
 *   dres_lo != NULL: dres is (dres_hi, dres_lo); dres must be NULL; dres_hi row-major or, dres_kb_rows > 0, K-blocked like x_hi (the T copy a row kernel
 *                    left for a ring GEMM is also the hi plane of the residual gradient).      dx_lo != NULL: the result is (dxT, dx_lo); dx32 may be NULL. */
int uia_layernorm_bwd3(void* stream, int dtype, int M, int D, int64_t ldx, const void* dy, const float* x, const void* x_hi, const int8_t* x_lo, int64_t x_kb_rows,
                       const float* gamma, float eps, const float* dres, const void* dres_hi, const int8_t* dres_lo, int64_t dres_kb_rows, float* dx32, void* dxT,
                       int8_t* dx_lo);

/* ---------------------------------------------------------------------------------------------
 * Mona adapter, all four variants (src/adapters/mona.py:75-487; equations SURVEY.md Appendix E.1).
 * The two projections run on uia_gemm, their weight gradients on uia_wgrad; these are the stages
 * in between.  bottleneck must be 64.
 *   pre_fwd      u = LN(x; norm.w, norm.b, eps)*gamma + x*gammax                  (mona.py:125,227,328,461)
 *   spatial_fwd  t [B,1+h*w,64] -> d = dropout(gelu([t_cls ; op(t_spatial)]))      (mona.py:129-147 + *MonaOp.forward)
 *   spatial_bwd  dd -> dt and += gradients of every adapter_conv parameter (fp32 atomics, caller zeroes)
 *   pre_bwd      dx = dy + du*gammax + LN'(du*gamma*norm.w); += d gamma, gammax, norm.w, norm.b
 * Dropout: keep_mask (uint8 [B,1+h*w,64]) if non-null, else the counter hash of (seed, element index)
 * when p_drop > 0; kept values are scaled by 1/(1-p_drop).  Pass the same seed / mask to fwd and bwd. */
typedef struct uia_mona_spatial_desc {
    int32_t variant, B, h, w, bott;
    const void* t;                /* T [B, 1+h*w, bott] output of project1 */
    void* d;                      /* T [B, 1+h*w, bott] forward output (input of project2) */
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *conv3_w, *conv3_b;   /* depth-wise 3x3 / 5x5 / 7x7 */
    const float *proj_w, *proj_b;                                             /* 1x1 projector [64,64] */
    const float* freq;                                                        /* freq_filter [64] (freq_enhanced, hybrid) */
    const float *ne1_w, *ne1_b, *ne3_w, *ne3_b;                               /* noise_estimator.{1,3} (noise_aware, hybrid) */
    float p_drop; uint64_t seed; const uint8_t* keep_mask;
    const void* dd;               /* backward: T grad wrt d */
    void* dt;                     /* backward: T grad wrt t */
    float *g_conv1_w, *g_conv1_b, *g_conv2_w, *g_conv2_b, *g_conv3_w, *g_conv3_b, *g_proj_w, *g_proj_b, *g_freq,
          *g_ne1_w, *g_ne1_b, *g_ne3_w, *g_ne3_b;
    float* ws;                    /* backward, optional: uia_mona_spatial_workspace_bytes(B) of scratch. With it every image writes one
                                     partial-gradient row and a second kernel adds the column sums into g_* in a fixed order
                                     (deterministic, no atomic contention); without it the kernel uses float atomics. */
} uia_mona_spatial_desc;
size_t uia_mona_spatial_workspace_bytes(int B);
int uia_mona_pre_fwd(void* stream, int dtype, int M, int D, const float* x, const float* norm_w, const float* norm_b,
                     const float* gamma, const float* gammax, float eps, void* u);
/* uia_mona_pre_fwd with project1 inside (bf16, D = 768, bottleneck 64): also writes t = u @ w1.T + b1 (mona.py:126-127), w1 = project1.weight [64, D] row-major,
 * t [M, ldt >= 64]; bit-identical to uia_mona_pre_fwd followed by uia_gemm on the N = 64 stream kernel. */
int uia_mona_pre_fwd_t(void* stream, int dtype, int M, int D, const float* x, const float* norm_w, const float* norm_b, const float* gamma, const float* gammax,
                       float eps, void* u, const void* w1, int64_t ldw1, const float* b1, void* t, int64_t ldt);
int uia_mona_pre_bwd(void* stream, int dtype, int M, int D, const void* du, const float* x, const float* dy,
                     const float* norm_w, const float* norm_b, const float* gamma, const float* gammax, float eps,
                     float* dx32, void* dxT, float* g_gamma, float* g_gammax, float* g_norm_w, float* g_norm_b, float* ws,
                     int64_t dxT_kb_rows);
/* dxT_kb_rows > 0: the T copy of dx is written K-blocked, [D/g][dxT_kb_rows][g] with g = 64 / sizeof(T) elements (uia_gemm_desc.a_kb_rows):
 * it is the A operand of the preceding block's fc2 data-gradient GEMM and of nothing else.
 * ws: caller-owned scratch of uia_mona_pre_bwd_workspace_bytes(M, D) bytes (per-workgroup partial rows of the four parameter
 * gradients, summed by a second small launch: a direct atomic add from every workgroup serialises on the same 4*D addresses) */
/* The same with project1's data gradient inside (bf16, bottleneck 64, D % 64 == 0, D <= 768): du = dt @ W1 is computed per 16-row tile on the matrix cores
 * from dt [M, 64] (the gradient wrt project1's output, mona.py:127) and w1t = project1.weight transposed, [D, 64] row-major, instead of being read
 * from a [M, D] tensor a K = 64 GEMM launch wrote; identical results (same products, same bf16 rounding of du). */
int uia_mona_pre_bwd_du(void* stream, int dtype, int M, int D, const void* dt, int64_t ldt, const void* w1t, int64_t ldw1, const float* x, const float* dy,
                        const float* norm_w, const float* norm_b, const float* gamma, const float* gammax, float eps,
                        float* dx32, void* dxT, float* g_gamma, float* g_gammax, float* g_norm_w, float* g_norm_b, float* ws, int64_t dxT_kb_rows);
/* uia_mona_pre_bwd_du on THREE-BYTE residual gradients (the form of uia_gemm_desc.resid_lo8; reference: the fp32 gradient autograd hands mona.py:151's residual add):
 * dy arrives as (dy_hi bf16 row-major [M, D], dy_lo int8 [M, D]) — what uia_layernorm_bwd3 left as (dxT, dx_lo) — and dx leaves as (dxT, dx_lo), dxT row-major or,
 * dxT_kb_rows > 0, K-blocked: 3 bytes read and 3 written per element of the stream instead of 4 and 4 + 2.  Same arithmetic as uia_mona_pre_bwd_du on the decoded values. */
int uia_mona_pre_bwd_du3(void* stream, int dtype, int M, int D, const void* dt, int64_t ldt, const void* w1t, int64_t ldw1, const float* x, const void* dy_hi, const int8_t* dy_lo,
                         const float* norm_w, const float* norm_b, const float* gamma, const float* gammax, float eps, void* dxT, int8_t* dx_lo, float* g_gamma,
                         float* g_gammax, float* g_norm_w, float* g_norm_b, float* ws, int64_t dxT_kb_rows);
size_t uia_mona_pre_bwd_workspace_bytes(int M, int D);
int uia_mona_spatial_fwd(void* stream, int dtype, const uia_mona_spatial_desc* d);
int uia_mona_spatial_bwd(void* stream, int dtype, const uia_mona_spatial_desc* d);

/* The WHOLE adapter forward of mona.py:319-362 (and :96-151, :198-253, :427-487) in one launch, one workgroup per image:
 *     y = x + project2(drop(gelu(spatial(project1(LN(x)·gamma + x·gammax)))))
 * pre -> project1 (768 -> 64 on MFMA, W1 resident in LDS) -> the spatial op on the fp32 [tokens][64] LDS tile -> GELU / dropout ->
 * project2 (64 -> 768 on MFMA, W2 fragments in registers) + residual, with u, t and d never travelling through HBM between stages.
 * bf16 only, bottleneck 64, grid width 14 or 4, D % 128 == 0, D <= 768 (uia_mona_fused_supported says whether a shape qualifies; everything
 * else runs as uia_mona_pre_fwd -> uia_gemm -> uia_mona_spatial_fwd -> uia_gemm with identical semantics).
 *   sp      variant, B, h, w, bott, the adapter_conv parameters, p_drop / seed / keep_mask;  sp.d (optional): T [B,1+hw,64] copy of d for the
 *           backward's weight gradient;  sp.t, sp.dd, sp.dt, sp.g_*, sp.ws are ignored
 *   x       fp32 [B,1+hw,D];  norm_w/norm_b/gamma/gammax fp32 [D];  w1 T [64,D] (project1.weight), b1 fp32 [64];  w2 T [D,64] (project2.weight), b2 fp32 [D]
 *   y32     fp32 [B,1+hw,D];  yT (optional) its T copy: row-major, or K-blocked with yT_kb_rows > 0 (uia_gemm_desc.a_kb_rows);
 *   rowsum_out (optional) receives (Σ, Σ²) of every row of y in the fixed-point form of uia_gemm_desc.rowsum_out — STORED, not added
 *           (an image's rows belong to one workgroup); ln_flag as in uia_gemm_desc
 *   u_out, t_out (optional) T [B·(1+hw), D] / [B,1+hw,64]: u and t for the backward (uia_wgrad of project1 / uia_mona_spatial_bwd) */
typedef struct uia_mona_fused_desc {
    uia_mona_spatial_desc sp;
    int32_t D; float eps;
    const float* x;
    const float *norm_w, *norm_b, *gamma, *gammax;
    const void* w1; const float* b1;
    const void* w2; const float* b2;
    float* y32;
    void* yT; int64_t yT_kb_rows;
    int64_t* rowsum_out;
    int32_t* ln_flag;
    void* u_out;
    void* t_out;
} uia_mona_fused_desc;
int uia_mona_fused_supported(int dtype, int D, int h, int w, int bott);
int uia_mona_fused_fwd(void* stream, int dtype, const uia_mona_fused_desc* d);

/* ---- LoRA rank update of a data gradient, up to three sources in one pass (csrc/lora_rank.hip) ----
 * Replaces, in the backward of the reference's LinearLoRA (src/adapters/lora.py:78-90: result += dropout(x) @ A.T @ B.T * scaling, one nn.Dropout per
 * wrapped Linear) applied to q, k and v of an OpenAI-CLIP block (inject_lora_to_clip, lora.py:115-199), the three read-modify-write launches
 *     dh += mask_i * (alpha * Q_i @ W_i.T) / (1 - p)        Q_i = dy_i @ B_i  [M, 64] (rank zero-padded to 64),  W_i = A_i.T  [N, 64]
 * by ONE pass over dh.  bf16; N % 256 == 0 and nsrc * N/4 * 144 bytes <= 160 KB (N <= 1024 with three sources).
 *   Q        bf16 [nsrc][M][ldq >= 64]; source s starts q_stride elements after source s-1
 *   W[s]     bf16 [N][ldw >= 64]
 *   out      bf16 [M][ldo >= N], read and written in place
 *   drop_p   0 = no dropout; otherwise mask_s is drawn per 8 output columns from (seed[s], (m*N + n) / 8) exactly as uia_dropout /
 *            uia_gemm_desc.drop_where = 2 draw it for an [M, N] tensor, so the forward's masks are reproduced from the seeds alone */
typedef struct uia_lora_rank_desc {
    int32_t M, N, nsrc;
    float alpha;
    const void* Q; int64_t ldq, q_stride;
    const void* W[3]; int64_t ldw;
    void* out; int64_t ldo;
    float drop_p;
    uint64_t seed[3];
} uia_lora_rank_desc;
int uia_lora_rank_update(void* stream, int dtype, const uia_lora_rank_desc* d);

/* LayerNorm + the down-projections of up to three LinearLoRA wrappers that share the input (q, k, v of an OpenAI-CLIP block; reference lora.py:82-87:
 * dropout(x) @ A.T, one nn.Dropout per wrapper) in one launch:  h = LayerNorm(x) (written, bf16 [M, D]) and  T[s] = dropout_s(h) @ A[s][:16].T.
 * bf16, D = 768 or 1024, rank <= 16 (A[s]: bf16 [>= 16, lda >= D], the first 16 rows are used); T: bf16 [nsrc][M][64] (t_stride elements between
 * sources), columns 0..15 hold the products and 16..63 zeros (the K-extension operand of uia_gemm_desc.A2 is 64 wide).  drop_p = 0: no dropout; otherwise
 * mask_s is the mask uia_gemm_desc.drop_where = 1 / uia_dropout draw for the [M, D] tensor h from seed[s] — uia_wgrad_drop regenerates it in the backward. */
typedef struct uia_ln_lora_desc {
    int32_t M, D, nsrc;
    float eps;
    const float* x; int64_t ldx;
    const float *gamma, *beta;
    void* h;
    const void* A[3]; int64_t lda;
    void* T; int64_t t_stride;
    float drop_p;
    uint64_t seed[3];
} uia_ln_lora_desc;
int uia_ln_lora_down(void* stream, int dtype, const uia_ln_lora_desc* d);

/* ---------------------------------------------------------------------------------------------
 * Task heads of the feature-pyramid adapter (reference src/third_party/timm/clip_adapter.py:47-57, 118-160).
 * uia_upsample_bilinear_fwd: nn.Upsample((H,W), mode="bilinear", align_corners=False) of a token-major map
 *   src[(b*h*w + y*w + x)*ld + c] -> dst[b][c][Y][X] (NCHW fp32).  _bwd: dsrc (token-major, overwritten) from ddst (NCHW),
 *   as a gather (fixed summation order).  The reference's Conv1x1 after the upsample is applied BEFORE it by the caller
 *   (it commutes with the interpolation), so C here is the number of classes.
 * uia_segment_mean_fwd: out[b][c] = mean_i x[(b*n + i)*ld + c]  (AdaptiveAvgPool2d(1) + Flatten); _bwd broadcasts dout/n. */
int uia_upsample_bilinear_fwd(void* stream, int B, int C, int h, int w, int H, int W, const float* src, int64_t ld, float* dst);
int uia_upsample_bilinear_bwd(void* stream, int B, int C, int h, int w, int H, int W, const float* ddst, float* dsrc, int64_t ld);
int uia_segment_mean_fwd(void* stream, int B, int n, int C, const float* x, int64_t ld, float* out);
int uia_segment_mean_bwd(void* stream, int B, int n, int C, const float* dout, float* dx, int64_t ld);
/* MONAI DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr, smooth_dr) of the segmentation entry points
 * (reference src/models/clipseg/segmentation.py:84, biomedclip/segmentation.py:75): loss (one float) and d loss / d logits in
 * one call.  logits, dlogits fp32 [B,C,H*W] (NCHW), label fp32 [B,H*W] holding class indices, 2 <= C <= 8;
 * ws: uia_dicece_workspace_bytes(B) of scratch. */
size_t uia_dicece_workspace_bytes(int B);
int uia_dicece_fwd_bwd(void* stream, int B, int C, int HW, const float* logits, const float* label, float smooth_nr, float smooth_dr,
                       float* ws, float* loss, float* dlogits);

/* ---------------------------------------------------------------------------------------------
 * Layout helpers around the GEMMs. */
int uia_cast(void* stream, int dtype, size_t n, const float* src, void* dst, float scale);          /* dst = T(scale*src) */
int uia_transpose_cast(void* stream, int dtype, int rows, int cols, const float* src, void* dst);   /* dst[c][r] = T(src[r][c]) */

/* All GEMM-operand forms of a set of small trainable fp32 matrices in ONE launch (the adapter weights after an optimiser step):
 * for matrix i (src [rows][cols] fp32), any non-null of
 *     row    [rows][cols]            T copy                         (weight [N = rows][K = cols] of y = x·Wᵀ)
 *     row_kb [cols/g][rows][g]       its K-blocked twin, g = 64 bytes of T elements (cols % g == 0)
 *     tr     [cols][rows]            T transpose                    (the dgrad weight)
 *     tr_kb  [rows/g][cols][g]       K-blocked twin of the transpose (rows % g == 0)
 * The descriptor table lives in DEVICE memory (the caller uploads it once and re-uses it every step).  Replaces, per matrix and
 * step, the reference's implicit autocast casts (torch.autocast around F.linear: /root/reference/src/models/biomedclip/finetune.py:277). */
typedef struct uia_pack_desc {
    const float* src;
    void* row;
    void* row_kb;
    void* tr;
    void* tr_kb;
    int32_t rows, cols;
    /* The destinations may be LARGER than the source: rows_pad x cols_pad (0 = rows / cols).  Only the source's elements are written, so
     * destinations zeroed once stay zero-padded — a LoRA factor [r, in] / [out, r] (r = 16) becomes the 64-wide GEMM operand in the same
     * launch that casts it.  scale (0 = 1) multiplies every element before the cast (lora.py:87's scaling folded into B). */
    int32_t rows_pad, cols_pad;
    float scale;
    int32_t reserved_;
} uia_pack_desc;
int uia_pack_weights(void* stream, int dtype, int n, const uia_pack_desc* descs_device, int max_elems);
int uia_im2col(void* stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out); /* model.py:221,234 */
/* same for any patch size (P need not be a multiple of 4), rows padded with zeros to ldo >= C*P*P columns (ViT-L/14: 588 -> 640) */
int uia_im2col_padded(void* stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* cols, int64_t ldo);
int uia_fill_cls(void* stream, int B, int N, int D, const float* cls, const float* pos0, float* x);   /* model.py:237-245 */
/* table [vocab, D], pos [max_pos, D]: L > max_pos is rejected; an id outside [0, vocab) (nn.Embedding raises for it) yields a NaN row,
 * which trips the training loops' non-finite-loss check instead of reading past the table */
int uia_embed(void* stream, int rows, int L, int D, int vocab, int max_pos, const int64_t* ids, const float* table, const float* pos,
              const float* type0, float* out);                                                          /* model.py:362-364 */
/* un-padded text tower (opt-in): rows are the valid tokens only; pos_idx[r] is the token's position inside its caption */
int uia_embed_packed(void* stream, int rows, int D, int vocab, int max_pos, const int64_t* ids, const int64_t* pos_idx, const float* table,
                     const float* pos, const float* type0, float* out);
/* nn.Embedding backward for --method full --tune_text_encoder: dtable[ids[r]] += dx[r] (fp32 atomics into a caller-zeroed table);
 * rows whose id equals pad_id are skipped (padding_idx). */
int uia_embed_bwd(void* stream, int rows, int D, int vocab, const int64_t* ids, const float* dx, float* dtable, int64_t pad_id);
int uia_gather_rows(void* stream, int n, int D, const float* src, const int64_t* idx, float* dst);  /* model.py:372 */
/* dst = (accumulate ? dst : 0) + src*keep/(1-p), keep from the counter hash of (seed, index): LoRA input dropout
 * (src/adapters/lora.py:82-83) and its backward (same seed). */
int uia_dropout(void* stream, int dtype, size_t n, const void* src, void* dst, float p, uint64_t seed, int accumulate);
/* out[N] += column sums of A[M,N] (bias gradients; LinearLoRA biases train: SURVEY Appendix C-4). */
int uia_colsum(void* stream, int dtype, int M, int N, const void* A, int64_t lda, float* out);

/* ---------------------------------------------------------------------------------------------
 * CLIPSeg decoder pieces (src/third_party/openai_clip/clipseg_adapter.py:73-98 → transformers CLIPSegDecoder [third-party],
 * SURVEY Appendix A.3).  Its Linear / convolution contractions are uia_gemm + uia_wgrad; uia_attn_* accepts head dim 16 / 32
 * (plain-VALU path) besides 64.  These are the remaining stages; the decoder is trainable, so all have backwards. */
int uia_layernorm_bwd_affine(void* stream, int dtype, int M, int D, const void* dy, const float* x, const float* gamma, float eps,
                             const float* dres, float* dx32, float* g_gamma, float* g_beta);   /* dx and += dgamma, dbeta */
int uia_film_fwd(void* stream, int B, int N, int C, const float* x, const float* mul, const float* add, float* y);   /* y = mul[b]*x + add[b] */
int uia_film_bwd(void* stream, int B, int N, int C, const float* dy, const float* x, const float* mul, float* dx, float* dmul, float* dadd);
/* 3x3 / pad-1 patches of the token grid: cols[(b*h*w+p)][(ky*3+kx)*C + c] = x[b][tok_off + nbr][c]; col2im is its adjoint
 * (rows < tok_off of dx are zeroed). */
int uia_im2col3x3(void* stream, int dtype, int B, int h, int w, int C, int ntok, int tok_off, const float* x, void* cols);
int uia_col2im3x3(void* stream, int dtype, int B, int h, int w, int C, int ntok, int tok_off, const void* dcols, float* dx);
/* two stacked kernel=stride transposed convolutions leave [B*h*w*k1*k1, >=k2*k2]; unshuffle writes logits [B, h*k1*k2, w*k1*k2] (+bias). */
int uia_unshuffle(void* stream, int dtype, int B, int h, int w, int k1, int k2, const void* tmp, int64_t ld, float bias, float* out);
int uia_shuffle(void* stream, int dtype, int B, int h, int w, int k1, int k2, const float* dout, void* dtmp, int64_t ld);
/* out = dy * act'(.) where y is the stored POST-activation for ReLU and the stored PRE-activation for GELU / QuickGELU */
int uia_act_bwd(void* stream, int dtype, size_t n, const void* dy, const void* y, int act, void* out);

/* ---------------------------------------------------------------------------------------------
 * Symmetric InfoNCE (src/losses/losses.py:23-47), forward + gradients of both feature matrices, fp32.
 * loss (1 float, device) is overwritten; dimg/dtxt (both or neither) receive grad_scale * dLoss/dfeat.
 * workspace: uia_infonce_workspace_bytes(B, E). */
size_t uia_infonce_workspace_bytes(int B, int E);
int uia_infonce_fwd_bwd(void* stream, int B, int E, const float* img, const float* txt, float inv_temp, float grad_scale,
                        float* loss, float* dimg, float* dtxt, void* workspace, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * One optimiser update on the flat fp32 adapter buffer: clip_grad_norm_(max_norm) + AdamW
 * (src/models/biomedclip/finetune.py:244-249,297-302).  g is read as grad_scale*g (1/world after the
 * all-reduce SUM).  ws2: 2 floats of device scratch; ws2[0] returns the squared gradient norm. */
int uia_adamw_clip_step(void* stream, size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float max_norm, int step, float grad_scale, float* ws2);

/* Guarded forms for loops that never read the loss on the host (round 5).  The reference decides per micro-batch on the host
 * (src/models/biomedclip/finetune.py:281-285: a non-finite loss skips the backward AND the update check of that iteration);
 * these take the same two decisions on the device.
 *   uia_grad_accum_guarded: mb[0..n) is the staging buffer one micro-batch's backward wrote; when *loss is finite it is added to
 *     acc[0..n), mb is zeroed either way.  acc has n + 4 floats: acc[n] = 0 (finite) / 1 (not) for THIS micro-batch — all-reduced
 *     with the gradients it becomes "ranks whose boundary micro-batch was non-finite".  stats[0] += loss (finite only);
 *     ctl (4 x int32): [0] updates applied, [1] micro-batches accumulated, [2] micro-batches skipped, [3] updates skipped;
 *     ok_log (optional): ok_log[log_index] = 1 / 0 for the host's end-of-epoch log.
 *   uia_adamw_clip_step_guarded: one clip + AdamW update from acc when acc[n] == 0, acc zeroed in the same pass (zero_grad);
 *     otherwise nothing but acc *= skip_scale (1/world under data parallelism: the summed buffer becomes this rank's share again).
 *     The update index t = ctl[0] lives on the device: lr = lr_min + (lr - lr_min)(1 + cos(pi t / t_max))/2 (CosineAnnealingLR,
 *     finetune.py:255; t_max = 0: constant lr) and the bias corrections use step t + 1, so a skipped update does not advance
 *     the schedule.  ws8: 8 floats of device scratch; ws8[0] returns the squared gradient norm, ws8[1] whether the update ran. */
int uia_grad_accum_guarded(void* stream, size_t n, float* acc, float* mb, const float* loss, float* stats, int32_t* ctl, uint8_t* ok_log, int64_t log_index);
int uia_adamw_clip_step_guarded(void* stream, size_t n, float* p, float* acc, float* m, float* v, float lr, float lr_min, int t_max, float beta1, float beta2,
                                float eps, float weight_decay, float max_norm, float grad_scale, float skip_scale, float* ws8, int32_t* ctl);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel exchange (new: the reference is single-process, finetune.py:287-302 accumulates
 * instead).  RCCL all-reduce on the caller's stream; the unique id travels through the host.
 * Every RCCL failure is reported as "rank r/world: <call> failed: <reason>" through uia_last_error(). */
int uia_comm_unique_id_bytes(void);
int uia_comm_get_unique_id(void* out, int bytes);
int uia_comm_init(int rank, int world, const void* unique_id, int bytes);
int uia_comm_world(void);
int uia_comm_initialised(void);   /* 1 once uia_comm_init succeeded (a one-rank communicator is valid: its all-reduce is the identity) */
int uia_allreduce_sum(void* stream, int dtype, void* buf, size_t n);
/* opt-in global-batch contrastive loss: recv[r*n_per_rank ...] = rank r's send buffer (RCCL all-gather). */
int uia_allgather(void* stream, int dtype, const void* send, void* recv, size_t n_per_rank);
int uia_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* UIA_HIP_H */
