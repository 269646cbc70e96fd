// uia_kernels.h — internal launcher declarations shared by the .hip files and capi.cpp.
// The public contract is include/uia_hip.h; nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/uia_hip.h"

typedef uia_gemm_desc UiaGemmParams;
typedef uia_attn_desc UiaAttnParams;

int uia_gemm_launch(hipStream_t stream, int dtype, const UiaGemmParams& p, int cfg);
int uia_attn_fwd_launch(hipStream_t stream, int dtype, const UiaAttnParams& p);
int uia_attn_bwd_launch(hipStream_t stream, int dtype, const UiaAttnParams& p, int cfg = 0);
int uia_layernorm_fwd_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const float* x, const float* gamma, const float* beta,
                             float eps, void* yT, float* y32, float* stats);
int uia_layernorm_bwd_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const void* dy, const float* x, const float* gamma, float eps,
                             const float* dres, float* dx32, void* dxT);
int uia_layernorm_bwd3_launch(hipStream_t stream, int dtype, int M, int D, long ldx, const void* dy, const float* x, const void* x_hi, const int8_t* x_lo,
                              long x_kb_rows, const float* gamma, float eps, const float* dres, const void* dres_hi, const int8_t* dres_lo, long dres_kb_rows,
                              float* dx32, void* dxT, int8_t* dx_lo);
int uia_lora_rank_update_launch(hipStream_t stream, int dtype, const uia_lora_rank_desc& p);
int uia_ln_lora_down_launch(hipStream_t stream, int dtype, const uia_ln_lora_desc& p);
int uia_cast_launch(hipStream_t stream, int dtype, size_t n, const float* src, void* dst, float scale);
int uia_transpose_cast_launch(hipStream_t stream, int dtype, int rows, int cols, const float* src, void* dst);
int uia_pack_weights_launch(hipStream_t stream, int dtype, int n, const uia_pack_desc* descs_device, int max_elems);
int uia_im2col_launch(hipStream_t stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out);
int uia_fill_cls_launch(hipStream_t stream, int B, int N, int D, const float* cls, const float* pos0, float* x);
int uia_embed_launch(hipStream_t stream, int rows, int L, int D, int vocab, int max_pos, const int64_t* ids, const float* table, const float* pos, const float* type0, float* out);
int uia_gather_rows_launch(hipStream_t stream, int n, int D, const float* src, const int64_t* idx, float* dst);
int uia_wgrad_group_launch(hipStream_t stream, int dtype, int n, int M, int I, int J, const void* const* A, long lda, const void* const* B, long ldb, float alpha,
                            float* const* dW, float* const* dbias, long ldw, int i_valid, int j_valid, float drop_p, const uint64_t* drop_seed, long drop_ld, int drop_col0);
int uia_wgrad_launch(hipStream_t stream, int dtype, int M, int I, int J, const void* A, long lda, const void* B, long ldb, float alpha, float* dW, float* dbias,
                     long ldw = 0, int i_valid = 0, int j_valid = 0, float drop_p = 0.f, uint64_t drop_seed = 0, long drop_ld = 0, int drop_col0 = 0);
int uia_mona_pre_fwd_launch(hipStream_t stream, int dtype, int M, int D, const float* x, const float* nw, const float* nb, const float* gamma,
                            const float* gammax, float eps, void* u);
int uia_mona_pre_fwd_t_launch(hipStream_t stream, int dtype, int M, int D, const float* x, const float* nw, const float* nb, const float* gamma,
                              const float* gammax, float eps, void* u, const void* w1, long ldw1, const float* b1, void* t, long ldt);
int uia_mona_pre_bwd_launch(hipStream_t stream, int dtype, int M, int D, const void* du, const float* x, const float* dy, const float* nw,
                            const float* nb, const float* gamma, const float* gammax, float eps, float* dx32, void* dxT, float* g_gamma,
                            float* g_gammax, float* g_nw, float* g_nb, float* ws, long dxT_kb_rows, const void* dt = nullptr, long ldt = 0,
                            const void* w1t = nullptr, long ldw1 = 0, const void* dy_hi = nullptr, const int8_t* dy_lo = nullptr, int8_t* dx_lo = nullptr);
size_t uia_mona_pre_bwd_ws_floats(int M, int D);
int uia_mona_spatial_fwd_launch(hipStream_t stream, int dtype, const uia_mona_spatial_desc& p);
int uia_mona_spatial_bwd_launch(hipStream_t stream, int dtype, const uia_mona_spatial_desc& p);
int uia_mona_fused_fwd_launch(hipStream_t stream, int dtype, const uia_mona_fused_desc& q);
size_t uia_infonce_workspace_floats(int B, int E);
int uia_infonce_launch(hipStream_t stream, int B, int E, const float* img, const float* txt, float inv_temp, float grad_scale, float* loss,
                       float* dimg, float* dtxt, float* ws, size_t ws_floats);
int uia_adamw_clip_launch(hipStream_t stream, size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float max_norm, int step, float grad_scale, float* ws);
int uia_grad_accum_guarded_launch(hipStream_t stream, size_t n, float* acc, float* mb, const float* loss, float* stats, int* ctl, unsigned char* ok_log, long log_index);
int uia_adamw_clip_guarded_launch(hipStream_t stream, size_t n, float* p, float* acc, float* m, float* v, float lr, float lr_min, int t_max, float beta1,
                                  float beta2, float eps, float weight_decay, float max_norm, float grad_scale, float skip_scale, float* ws8, int* ctl);
int uia_dropout_launch(hipStream_t stream, int dtype, size_t n, const void* src, void* dst, float p, uint64_t seed, int accumulate);
int uia_colsum_launch(hipStream_t stream, int dtype, int M, int N, const void* A, long lda, float* out);
int uia_attn_small_launch(hipStream_t stream, int dtype, const UiaAttnParams& p, bool bwd);
bool uia_attn_dh16_ok(int dtype, const UiaAttnParams& p);                            // attention_dh16.hip: bf16, head dim 16, no mask, L <= 512 on the matrix cores
int uia_attn_dh16_launch(hipStream_t stream, const UiaAttnParams& p, bool bwd);
int uia_layernorm_bwd_affine_launch(hipStream_t stream, int dtype, int M, int D, const void* dy, const float* x, const float* gamma, float eps,
                                    const float* dres, float* dx32, float* g_gamma, float* g_beta);
int uia_film_fwd_launch(hipStream_t stream, int B, int N, int C, const float* x, const float* mul, const float* add, float* y);
int uia_film_bwd_launch(hipStream_t stream, int B, int N, int C, const float* dy, const float* x, const float* mul, float* dx, float* dmul, float* dadd);
int uia_im2col3x3_launch(hipStream_t stream, int dtype, int B, int h, int w, int C, int ntok, int tok_off, const float* x, void* cols);
int uia_col2im3x3_launch(hipStream_t stream, int dtype, int B, int h, int w, int C, int ntok, int tok_off, const void* dcols, float* dx);
int uia_unshuffle_launch(hipStream_t stream, int dtype, int B, int h, int w, int k1, int k2, const void* tmp, long ld, float bias, float* out);
int uia_shuffle_launch(hipStream_t stream, int dtype, int B, int h, int w, int k1, int k2, const float* dout, void* dtmp, long ld);
int uia_act_bwd_launch(hipStream_t stream, int dtype, size_t n, const void* dy, const void* y, int act, void* out);
size_t uia_mona_spatial_ws_floats(int B);
int uia_upsample_bilinear_launch(hipStream_t stream, bool bwd, int B, int C, int h, int w, int H, int W, const float* in, float* out, long ld);
int uia_segment_mean_launch(hipStream_t stream, bool bwd, int B, int n, int C, const float* in, float* out, long ld);
size_t uia_dicece_ws_floats(int B);
int uia_dicece_launch(hipStream_t stream, int B, int C, int HW, const float* logits, const float* label, float nr, float dr, float* ws, float* loss, float* dlogits);
int uia_im2col_padded_launch(hipStream_t stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out, long ldo);
int uia_embed_bwd_launch(hipStream_t stream, int rows, int D, int vocab, const int64_t* ids, const float* dx, float* dtable, long pad_id);
int uia_embed_packed_launch(hipStream_t stream, int rows, int D, int vocab, int max_pos, const int64_t* ids, const int64_t* pos_idx, const float* table, const float* pos, const float* type0, float* out);
