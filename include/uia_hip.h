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
} uia_gemm_desc;
int uia_gemm(void* stream, int dtype, const uia_gemm_desc* d, int tile_cfg /* 0 = auto */);

/* Parameter gradient of a Linear:  dW[I,J] += Σ_m A[m,I]ᵀ·B[m,J]  and optionally dbias[I] += Σ_m A[m,I]
 * (fp32 accumulate with atomics into caller-zeroed buffers).  One of I, J must be ≤ 64·k.
 * Replaces autograd's wgrad of mona.py:127,148 and lora.py:87. */
int uia_wgrad(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb,
              float alpha, float* dW, float* dbias_A);

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
} uia_attn_desc;
int uia_attn_fwd(void* stream, int dtype, const uia_attn_desc* d);
int uia_attn_bwd(void* stream, int dtype, const uia_attn_desc* d);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over fp32 rows (model.py:163-169; timm / HF LayerNorm [third-party]).
 * x rows may be strided by ldx (elements); y / dy are compact [M,D].  Backward is for FROZEN
 * gamma/beta (dx only): dx = dres + LN'(dy); statistics are recomputed from x. */
int uia_layernorm_fwd(void* stream, int dtype, int M, int D, int64_t ldx, const float* x, const float* gamma, const float* beta,
                      float eps, void* yT, float* y32);
int uia_layernorm_bwd(void* stream, int dtype, int M, int D, int64_t ldx, const void* dy, const float* x, const float* gamma,
                      float eps, const float* dres, float* dx32, void* dxT);

/* ---------------------------------------------------------------------------------------------
 * Layout helpers around the GEMMs. */
int uia_cast(void* stream, int dtype, size_t n, const float* src, void* dst, float scale);          /* dst = T(scale*src) */
int uia_transpose_cast(void* stream, int dtype, int rows, int cols, const float* src, void* dst);   /* dst[c][r] = T(src[r][c]) */
int uia_im2col(void* stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out); /* model.py:221,234 */
int uia_fill_cls(void* stream, int B, int N, int D, const float* cls, const float* pos0, float* x);   /* model.py:237-245 */
int uia_embed(void* stream, int rows, int L, int D, const int64_t* ids, const float* table, const float* pos,
              const float* type0, float* out);                                                          /* model.py:362-364 */
int uia_gather_rows(void* stream, int n, int D, const float* src, const int64_t* idx, float* dst);  /* model.py:372 */

#ifdef __cplusplus
}
#endif
#endif /* UIA_HIP_H */
