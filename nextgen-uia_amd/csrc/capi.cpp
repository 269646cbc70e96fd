// capi.cpp — extern "C" entry points of libuia_hip.so (declared in include/uia_hip.h).
// Thin: null checks on the descriptor, then the typed launcher.  No allocation, no sync.
#include "uia_kernels.h"

void uia_set_error(const char* fmt, ...);
#define NEED(ptr, name) do { if (!(ptr)) { uia_set_error("%s: null descriptor", name); return -1; } } while (0)

extern "C" {

int uia_version(void) { return 100; }

int uia_gemm(void* stream, int dtype, const uia_gemm_desc* d, int tile_cfg) {
    NEED(d, "uia_gemm");
    return uia_gemm_launch((hipStream_t)stream, dtype, *d, tile_cfg);
}
int uia_lora_rank_update(void* stream, int dtype, const uia_lora_rank_desc* d) {
    NEED(d, "uia_lora_rank_update");
    return uia_lora_rank_update_launch((hipStream_t)stream, dtype, *d);
}
int uia_ln_lora_down(void* stream, int dtype, const uia_ln_lora_desc* d) {
    NEED(d, "uia_ln_lora_down");
    return uia_ln_lora_down_launch((hipStream_t)stream, dtype, *d);
}
int uia_wgrad(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb, float alpha, float* dW, float* dbias_A) {
    return uia_wgrad_launch((hipStream_t)stream, dtype, M, I, J, A, lda, B, ldb, alpha, dW, dbias_A);
}
int uia_wgrad_ex(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb, float alpha, float* dW, int64_t ldw,
                 int i_valid, int j_valid, float* dbias_A) {
    if (ldw <= 0) { uia_set_error("uia_wgrad_ex: ldw=%lld must be positive", (long long)ldw); return -1; }
    return uia_wgrad_launch((hipStream_t)stream, dtype, M, I, J, A, lda, B, ldb, alpha, dW, dbias_A, ldw, i_valid, j_valid);
}
int uia_wgrad_group(void* stream, int dtype, const uia_wgrad_group_desc* d) {
    if (!d) { uia_set_error("uia_wgrad_group: null descriptor"); return -1; }
    if (d->n < 1 || d->n > 4) { uia_set_error("uia_wgrad_group: n=%d outside 1..4", d->n); return -1; }
    if (d->ldw <= 0) { uia_set_error("uia_wgrad_group: ldw=%lld must be positive", (long long)d->ldw); return -1; }
    float* dW[4];
    float* db[4];
    for (int g = 0; g < 4; ++g) { dW[g] = d->dW[g]; db[g] = d->dbias_A[g]; }
    if (d->drop_p > 0.f) for (int g = 0; g < d->n; ++g) if (db[g]) { uia_set_error("uia_wgrad_group: dbias with dropout on B is not a form of the kernel"); return -1; }
    return uia_wgrad_group_launch((hipStream_t)stream, dtype, d->n, d->M, d->I, d->J, d->A, (long)d->lda, d->B, (long)d->ldb, d->alpha, dW, db, (long)d->ldw, d->i_valid, d->j_valid,
                                  d->drop_p, d->drop_seed, (long)d->drop_ld, d->drop_col0);
}
int uia_wgrad_drop(void* stream, int dtype, int M, int I, int J, const void* A, int64_t lda, const void* B, int64_t ldb, float alpha, float* dW, int64_t ldw,
                   int i_valid, int j_valid, float drop_p, uint64_t seed, int64_t drop_ld, int drop_col0) {
    if (ldw <= 0) { uia_set_error("uia_wgrad_drop: ldw=%lld must be positive", (long long)ldw); return -1; }
    if (!(drop_p > 0.f)) { uia_set_error("uia_wgrad_drop: drop_p=%f must be in (0, 1); uia_wgrad_ex is the form without dropout", (double)drop_p); return -1; }
    return uia_wgrad_launch((hipStream_t)stream, dtype, M, I, J, A, lda, B, ldb, alpha, dW, nullptr, ldw, i_valid, j_valid, drop_p, seed, drop_ld, drop_col0);
}
int uia_attn_fwd(void* stream, int dtype, const uia_attn_desc* d) {
    NEED(d, "uia_attn_fwd");
    return uia_attn_fwd_launch((hipStream_t)stream, dtype, *d);
}
int uia_attn_bwd(void* stream, int dtype, const uia_attn_desc* d) {
    NEED(d, "uia_attn_bwd");
    return uia_attn_bwd_launch((hipStream_t)stream, dtype, *d, 0);
}
int uia_attn_bwd_cfg(void* stream, int dtype, const uia_attn_desc* d, int cfg) {
    NEED(d, "uia_attn_bwd_cfg");
    return uia_attn_bwd_launch((hipStream_t)stream, dtype, *d, cfg);
}
int uia_layernorm_fwd(void* stream, int dtype, int M, int D, int64_t ldx, const float* x, const float* gamma, const float* beta, float eps, void* yT, float* y32) {
    return uia_layernorm_fwd_launch((hipStream_t)stream, dtype, M, D, ldx, x, gamma, beta, eps, yT, y32, nullptr);
}
int uia_layernorm_fwd_stats(void* stream, int dtype, int M, int D, int64_t ldx, const float* x, const float* gamma, const float* beta, float eps, void* yT, float* y32,
                            float* stats) {
    return uia_layernorm_fwd_launch((hipStream_t)stream, dtype, M, D, ldx, x, gamma, beta, eps, yT, y32, stats);
}
int uia_layernorm_bwd(void* stream, int dtype, int M, int D, int64_t ldx, const void* dy, const float* x, const float* gamma, float eps, const float* dres, float* dx32, void* dxT) {
    return uia_layernorm_bwd_launch((hipStream_t)stream, dtype, M, D, ldx, dy, x, gamma, eps, dres, dx32, dxT);
}
int uia_layernorm_bwd3(void* stream, int dtype, int M, int D, int64_t ldx, const void* dy, const float* x, const void* x_hi, const int8_t* x_lo, int64_t x_kb_rows,
                       const float* gamma, float eps, const float* dres, const void* dres_hi, const int8_t* dres_lo, int64_t dres_kb_rows, float* dx32, void* dxT,
                       int8_t* dx_lo) {
    return uia_layernorm_bwd3_launch((hipStream_t)stream, dtype, M, D, ldx, dy, x, x_hi, x_lo, (long)x_kb_rows, gamma, eps, dres, dres_hi, dres_lo, (long)dres_kb_rows, dx32, dxT,
                                     dx_lo);
}
int uia_cast(void* stream, int dtype, size_t n, const float* src, void* dst, float scale) { return uia_cast_launch((hipStream_t)stream, dtype, n, src, dst, scale); }
int uia_transpose_cast(void* stream, int dtype, int rows, int cols, const float* src, void* dst) { return uia_transpose_cast_launch((hipStream_t)stream, dtype, rows, cols, src, dst); }
int uia_pack_weights(void* stream, int dtype, int n, const uia_pack_desc* descs_device, int max_elems) { return uia_pack_weights_launch((hipStream_t)stream, dtype, n, descs_device, max_elems); }
int uia_im2col(void* stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* out) { return uia_im2col_launch((hipStream_t)stream, dtype, B, C, H, W, P, img, out); }
int uia_fill_cls(void* stream, int B, int N, int D, const float* cls, const float* pos0, float* x) { return uia_fill_cls_launch((hipStream_t)stream, B, N, D, cls, pos0, x); }
int uia_embed(void* stream, int rows, int L, int D, int vocab, int max_pos, const int64_t* ids, const float* table, const float* pos, const float* type0, float* out) {
    return uia_embed_launch((hipStream_t)stream, rows, L, D, vocab, max_pos, ids, table, pos, type0, out);
}
int uia_gather_rows(void* stream, int n, int D, const float* src, const int64_t* idx, float* dst) { return uia_gather_rows_launch((hipStream_t)stream, n, D, src, idx, dst); }

int uia_mona_pre_fwd(void* stream, int dtype, int M, int D, const float* x, const float* norm_w, const float* norm_b, const float* gamma,
                     const float* gammax, float eps, void* u) {
    return uia_mona_pre_fwd_launch((hipStream_t)stream, dtype, M, D, x, norm_w, norm_b, gamma, gammax, eps, u);
}
int uia_mona_pre_bwd(void* stream, int dtype, int M, int D, const void* du, const float* x, const float* dy, const float* norm_w,
                     const float* norm_b, const float* gamma, const float* gammax, float eps, float* dx32, void* dxT, float* g_gamma,
                     float* g_gammax, float* g_norm_w, float* g_norm_b, float* ws, int64_t dxT_kb_rows) {
    return uia_mona_pre_bwd_launch((hipStream_t)stream, dtype, M, D, du, x, dy, norm_w, norm_b, gamma, gammax, eps, dx32, dxT, g_gamma, g_gammax,
                                   g_norm_w, g_norm_b, ws, (long)dxT_kb_rows);
}
int uia_mona_pre_fwd_t(void* stream, int dtype, int M, int D, const float* x, const float* norm_w, const float* norm_b, const float* gamma, const float* gammax,
                       float eps, void* u, const void* w1, int64_t ldw1, const float* b1, void* t, int64_t ldt) {
    return uia_mona_pre_fwd_t_launch((hipStream_t)stream, dtype, M, D, x, norm_w, norm_b, gamma, gammax, eps, u, w1, (long)ldw1, b1, t, (long)ldt);
}
int uia_mona_pre_bwd_du(void* stream, int dtype, int M, int D, const void* dt, int64_t ldt, const void* w1t, int64_t ldw1, const float* x, const float* dy,
                        const float* norm_w, const float* norm_b, const float* gamma, const float* gammax, float eps, float* dx32, void* dxT, float* g_gamma,
                        float* g_gammax, float* g_norm_w, float* g_norm_b, float* ws, int64_t dxT_kb_rows) {
    if (!dt || !w1t) { uia_set_error("uia_mona_pre_bwd_du: null dt / w1t"); return -1; }
    return uia_mona_pre_bwd_launch((hipStream_t)stream, dtype, M, D, nullptr, x, dy, norm_w, norm_b, gamma, gammax, eps, dx32, dxT, g_gamma, g_gammax,
                                   g_norm_w, g_norm_b, ws, (long)dxT_kb_rows, dt, (long)ldt, w1t, (long)ldw1);
}
int uia_mona_pre_bwd_du3(void* stream, int dtype, int M, int D, const void* dt, int64_t ldt, const void* w1t, int64_t ldw1, const float* x, const void* dy_hi, const int8_t* dy_lo,
                         const float* norm_w, const float* norm_b, const float* gamma, const float* gammax, float eps, void* dxT, int8_t* dx_lo, float* g_gamma,
                         float* g_gammax, float* g_norm_w, float* g_norm_b, float* ws, int64_t dxT_kb_rows) {
    if (!dt || !w1t || !dy_hi || !dy_lo || !dx_lo) { uia_set_error("uia_mona_pre_bwd_du3: null dt / w1t / dy_hi / dy_lo / dx_lo"); return -1; }
    return uia_mona_pre_bwd_launch((hipStream_t)stream, dtype, M, D, nullptr, x, nullptr, norm_w, norm_b, gamma, gammax, eps, nullptr, dxT, g_gamma, g_gammax,
                                   g_norm_w, g_norm_b, ws, (long)dxT_kb_rows, dt, (long)ldt, w1t, (long)ldw1, dy_hi, dy_lo, dx_lo);
}
size_t uia_mona_pre_bwd_workspace_bytes(int M, int D) { return uia_mona_pre_bwd_ws_floats(M, D) * sizeof(float); }
int uia_mona_spatial_fwd(void* stream, int dtype, const uia_mona_spatial_desc* d) {
    NEED(d, "uia_mona_spatial_fwd");
    return uia_mona_spatial_fwd_launch((hipStream_t)stream, dtype, *d);
}
int uia_mona_spatial_bwd(void* stream, int dtype, const uia_mona_spatial_desc* d) {
    NEED(d, "uia_mona_spatial_bwd");
    return uia_mona_spatial_bwd_launch((hipStream_t)stream, dtype, *d);
}
int uia_mona_fused_fwd(void* stream, int dtype, const uia_mona_fused_desc* d) {
    NEED(d, "uia_mona_fused_fwd");
    return uia_mona_fused_fwd_launch((hipStream_t)stream, dtype, *d);
}

size_t uia_infonce_workspace_bytes(int B, int E) { return uia_infonce_workspace_floats(B, E) * sizeof(float); }
int uia_infonce_fwd_bwd(void* stream, int B, int E, const float* img, const float* txt, float inv_temp, float grad_scale, float* loss,
                        float* dimg, float* dtxt, void* workspace, size_t workspace_bytes) {
    return uia_infonce_launch((hipStream_t)stream, B, E, img, txt, inv_temp, grad_scale, loss, dimg, dtxt, (float*)workspace, workspace_bytes / sizeof(float));
}
int uia_adamw_clip_step(void* stream, size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float max_norm, int step, float grad_scale, float* ws2) {
    return uia_adamw_clip_launch((hipStream_t)stream, n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, max_norm, step, grad_scale, ws2);
}
int uia_grad_accum_guarded(void* stream, size_t n, float* acc, float* mb, const float* loss, float* stats, int32_t* ctl, uint8_t* ok_log, int64_t log_index) {
    return uia_grad_accum_guarded_launch((hipStream_t)stream, n, acc, mb, loss, stats, (int*)ctl, ok_log, (long)log_index);
}
int uia_adamw_clip_step_guarded(void* stream, size_t n, float* p, float* acc, float* m, float* v, float lr, float lr_min, int t_max, float beta1, float beta2,
                                float eps, float weight_decay, float max_norm, float grad_scale, float skip_scale, float* ws8, int32_t* ctl) {
    return uia_adamw_clip_guarded_launch((hipStream_t)stream, n, p, acc, m, v, lr, lr_min, t_max, beta1, beta2, eps, weight_decay, max_norm, grad_scale, skip_scale, ws8, (int*)ctl);
}

int uia_dropout(void* stream, int dtype, size_t n, const void* src, void* dst, float p, uint64_t seed, int accumulate) {
    return uia_dropout_launch((hipStream_t)stream, dtype, n, src, dst, p, seed, accumulate);
}
int uia_colsum(void* stream, int dtype, int M, int N, const void* A, int64_t lda, float* out) {
    return uia_colsum_launch((hipStream_t)stream, dtype, M, N, A, lda, out);
}

int uia_layernorm_bwd_affine(void* stream, int dtype, int M, int D, const void* dy, const float* x, const float* gamma, float eps, const float* dres,
                             float* dx32, float* g_gamma, float* g_beta) {
    return uia_layernorm_bwd_affine_launch((hipStream_t)stream, dtype, M, D, dy, x, gamma, eps, dres, dx32, g_gamma, g_beta);
}
int uia_film_fwd(void* stream, int B, int N, int C, const float* x, const float* mul, const float* add, float* y) { return uia_film_fwd_launch((hipStream_t)stream, B, N, C, x, mul, add, y); }
int uia_film_bwd(void* stream, int B, int N, int C, const float* dy, const float* x, const float* mul, float* dx, float* dmul, float* dadd) {
    return uia_film_bwd_launch((hipStream_t)stream, B, N, C, dy, x, mul, dx, dmul, dadd);
}
int uia_im2col3x3(void* stream, int dtype, int B, int h, int w, int C, int ntok, int tok_off, const float* x, void* cols) {
    return uia_im2col3x3_launch((hipStream_t)stream, dtype, B, h, w, C, ntok, tok_off, x, cols);
}
int uia_col2im3x3(void* stream, int dtype, int B, int h, int w, int C, int ntok, int tok_off, const void* dcols, float* dx) {
    return uia_col2im3x3_launch((hipStream_t)stream, dtype, B, h, w, C, ntok, tok_off, dcols, dx);
}
int uia_unshuffle(void* stream, int dtype, int B, int h, int w, int k1, int k2, const void* tmp, int64_t ld, float bias, float* out) {
    return uia_unshuffle_launch((hipStream_t)stream, dtype, B, h, w, k1, k2, tmp, ld, bias, out);
}
int uia_shuffle(void* stream, int dtype, int B, int h, int w, int k1, int k2, const float* dout, void* dtmp, int64_t ld) {
    return uia_shuffle_launch((hipStream_t)stream, dtype, B, h, w, k1, k2, dout, dtmp, ld);
}
int uia_act_bwd(void* stream, int dtype, size_t n, const void* dy, const void* y, int act, void* out) { return uia_act_bwd_launch((hipStream_t)stream, dtype, n, dy, y, act, out); }

size_t uia_mona_spatial_workspace_bytes(int B) { return uia_mona_spatial_ws_floats(B) * sizeof(float); }

int uia_upsample_bilinear_fwd(void* stream, int B, int C, int h, int w, int H, int W, const float* src, int64_t ld, float* dst) {
    return uia_upsample_bilinear_launch((hipStream_t)stream, false, B, C, h, w, H, W, src, dst, (long)ld);
}
int uia_upsample_bilinear_bwd(void* stream, int B, int C, int h, int w, int H, int W, const float* ddst, float* dsrc, int64_t ld) {
    return uia_upsample_bilinear_launch((hipStream_t)stream, true, B, C, h, w, H, W, ddst, dsrc, (long)ld);
}
int uia_segment_mean_fwd(void* stream, int B, int n, int C, const float* x, int64_t ld, float* out) {
    return uia_segment_mean_launch((hipStream_t)stream, false, B, n, C, x, out, (long)ld);
}
int uia_segment_mean_bwd(void* stream, int B, int n, int C, const float* dout, float* dx, int64_t ld) {
    return uia_segment_mean_launch((hipStream_t)stream, true, B, n, C, dout, dx, (long)ld);
}

size_t uia_dicece_workspace_bytes(int B) { return uia_dicece_ws_floats(B) * sizeof(float); }
int uia_dicece_fwd_bwd(void* stream, int B, int C, int HW, const float* logits, const float* label, float smooth_nr, float smooth_dr,
                       float* ws, float* loss, float* dlogits) {
    return uia_dicece_launch((hipStream_t)stream, B, C, HW, logits, label, smooth_nr, smooth_dr, ws, loss, dlogits);
}

int uia_im2col_padded(void* stream, int dtype, int B, int C, int H, int W, int P, const float* img, void* cols, int64_t ldo) {
    return uia_im2col_padded_launch((hipStream_t)stream, dtype, B, C, H, W, P, img, cols, (long)ldo);
}

int uia_embed_bwd(void* stream, int rows, int D, int vocab, const int64_t* ids, const float* dx, float* dtable, int64_t pad_id) {
    return uia_embed_bwd_launch((hipStream_t)stream, rows, D, vocab, ids, dx, dtable, (long)pad_id);
}

int uia_embed_packed(void* stream, int rows, int D, int vocab, int max_pos, const int64_t* ids, const int64_t* pos_idx, const float* table,
                     const float* pos, const float* type0, float* out) {
    return uia_embed_packed_launch((hipStream_t)stream, rows, D, vocab, max_pos, ids, pos_idx, table, pos, type0, out);
}

}  // extern "C"
