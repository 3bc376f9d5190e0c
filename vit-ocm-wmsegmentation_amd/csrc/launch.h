// launch.h — host-side launchers shared between the kernel translation units and the
// engine (engine.hip). Every launcher only enqueues work on `s`; none synchronises.
#pragma once
#include "common.h"

// ---- kernels_misc.hip
// out_kind: 0 fp32, 1 bf16, 2 split-bf16 pairs
hipError_t launch_layernorm(const float *x, const float *gamma, const float *beta, void *y, int out_kind,
                            int64_t rows, int dim, float eps, hipStream_t s);
static inline int ln_kind_of_prec(int prec) { return prec == 0 ? 1 : prec == 1 ? 0 : 2; }
hipError_t launch_cast_bf16(const float *src, bf16 *dst, size_t count, hipStream_t s);
hipError_t launch_cast_split(const float *src, void *dst, size_t count, hipStream_t s);
hipError_t launch_merge_split(const void *src, float *dst, size_t count, hipStream_t s);
hipError_t launch_fold_split(const float *src, void *dst, int D, int C, int pp, hipStream_t s);
// sum over the channel axis of a (D, C, p, p) conv weight -> bf16 (D, p*p): grayscale fold.
hipError_t launch_fold_cast_bf16(const float *src, bf16 *dst, int D, int C, int pp, hipStream_t s);
hipError_t launch_fold_f32(const float *src, float *dst, int D, int C, int pp, hipStream_t s);
hipError_t launch_im2col3x3(int prec, const float *in, void *out, int batch, int h, int w, int C, int relu, hipStream_t s);
hipError_t launch_cls_rows(const float *cls, const float *pos, float *x, int batch, int n_tokens, int dim,
                           hipStream_t s);
// ---- LayerNorm folded into its consumer GEMM (split-bf16 forward) ----
// What a consuming epilogue needs to finish LN(x) W^T + b from the raw products acc = x W'^T (W' = W * gamma):
//   out = rstd * (acc - mu * c) + d,  mu = S1 / dim, rstd = 1 / sqrt(S2 / dim - mu^2 + eps)
// The sums of a row live in dim / 64 slots, one per 64 columns: a producing tile writes (sum, sum of squares) over ITS
// columns into its first slot and zeros into the other slots it covers, the consumer adds the slots in index order. Every
// slot is written exactly once per layer, so nothing is zeroed beforehand and no atomics are involved: the statistics —
// and with them every output — are the same bits from run to run, whatever the tile shapes on either side.
struct LnFold {
    const float *stats = nullptr;  // [rows][dim / 64][2] of the consumer's A rows, or null: no folding
    const float *c = nullptr, *d = nullptr;  // per output column
    float inv_dim = 0.f, eps = 0.f;
    int nslot = 0;  // dim / 64
};
// Row centring (round 4). LayerNorm does not see a constant added to a whole row, but the split-bf16 rounding of the operand
// does: a row whose mean is 30 standard deviations from zero carries 30 x the rounding error of its centred self into the
// consumer's products (measured: attention L_inf 5e-5 -> 5e-3 on such rows). So the pairs a producer writes are those of
// x - s[row], and the sums those of x - s[row], with s[row] a per-row constant that every column tile of the row can know without
// talking to the others: the row's mean AT THE PREVIOUS LayerNorm site (previous shift + previous sums / N; the residual stream
// moves little between sites) — at the first site a per-token constant (mean of the positional row + mean of the bias; the
// cls row: its exact mean). The consumer is unchanged: LN(x - s) = LN(x). x itself is stored unshifted.
// What a producing epilogue (x = resid + acc) writes besides x when the next LayerNorm is folded
struct StatsOut {
    void *xs = nullptr;      // [rows][N] split pairs of x - shift[row], or null
    float *stats = nullptr;  // [rows][N / 64][2] sums of x - shift[row]
    float *shift = nullptr;  // [rows] out: the constant subtracted from each row (null: nothing is subtracted)
    const float *prev_stats = nullptr, *prev_shift = nullptr;  // the previous site's sums and shifts of the same rows, or
    const float *tok_shift = nullptr;                          // (first site) [n_tokens] per-token constants
    // Few rows and a long contraction (mlp.fc2 of a one-tile-per-call forward: M = 197, K = 1536 on 24 workgroups): the K range
    // is cut into OCM_SPLITK slices, one workgroup per (tile, slice) writes fp32 partial sums here ([slices][M][N]) and a row
    // kernel adds them in slice order with bias and residual (and produces xs / stats). Null: never split.
    float *part = nullptr;
};
constexpr int OCM_SPLITK = 4, OCM_SPLITK_MAX_ROWS = 512;  // at 2305 rows (one ViT-S/8 window) the split form measures the same as the plain kernel
// x = resid + bias + sum over slices of part[slice] (fixed order); optionally split pairs and row sums of x
hipError_t launch_splitk_finish(const float *part, int slices, const float *bias, const float *resid, float *x, void *xs,
                                float *stats, int M, int N, hipStream_t s, const StatsOut &so = StatsOut());
// + (shift / tok_shift non-null) the cls rows' exact means into shift[b * n_tokens] and the per-token constants
// tok_shift[n] = mean(pos[n]) + mean(bias) of the patch rows, which the patch-embedding epilogue subtracts
hipError_t launch_cls_rows_stats(const float *cls, const float *pos, float *x, void *xs, float *stats, int batch,
                                 int n_tokens, int dim, hipStream_t s, float *shift = nullptr, float *tok_shift = nullptr,
                                 const float *pe_bias = nullptr);
hipError_t launch_fold_ln(const float *W, const float *gamma, const float *beta, const float *bias, void *Wf, float *cvec,
                          float *dvec, int N, int K, hipStream_t s);

// `prec` selects the operand element type of the contraction kernels: 0 = bf16 (OCM_PREC_BF16),
// 1 = fp32 (OCM_PREC_FP32), 2 = split-bf16 pairs (OCM_PREC_BF16X3, common.h: sp32). Activation buffers
// (a, q, k, vt, ctx, ...) and weights are of that type.
// ---- kernels_gemm.hip
hipError_t launch_linear(int prec, const void *a, const void *w, const float *bias, const float *resid, void *out, int M,
                         int N, int K, int epilogue, hipStream_t s, const LnFold &ln = LnFold(),
                         const StatsOut &so = StatsOut());
// head_dim 64: q / k / vt are the attention kernels' operand copies (vt may be null when V is never read: want_v
// false skips the V third). Any other head_dim (multiple of 8): q = k = vt = null, qkv_f32 (3,B,H,N,hd) is the output.
// proj / fc2 + residual fused with the LayerNorm that follows (full-row tiles, D in {128, 256, 384})
bool linear_resid_ln_supported(int D);
hipError_t launch_linear_resid_ln(int prec, const void *a, const void *w, const float *bias, const float *resid, float *x,
                                  const float *gamma, const float *beta, void *xn, int M, int D, int K, float eps,
                                  hipStream_t s);
hipError_t launch_qkv(int prec, const void *a, const void *w, const float *bias, void *q, void *k, void *vt,
                      float *qkv_f32, int batch, int n_tokens, int n_pad, int heads, int head_dim, bool want_v,
                      hipStream_t s, const LnFold &ln = LnFold());
// Attention for heads that are not 64 wide (model.py:96-97: 3 heads x 128): fp32 FMA arithmetic on the fp32 qkv tensor.
// ctx (activation type of `prec`, [B*N][H*hd]) / attn (B,H,N,N) / rows (B,H,n_rows,N-1) are optional outputs.
hipError_t launch_attention_generic(int prec, const float *qkv, void *ctx, float *attn, const int32_t *query_rows,
                                    int n_rows, float *rows, int batch, int n_tokens, int heads, int head_dim, float scale,
                                    hipStream_t s);
struct PatchArgs {
    const float *image;
    int64_t sb, sc, sy;
    const int32_t *origins;
    int batch, hp, wp, p, chans;  // hp x wp patches per tile
    const float *mask = nullptr;      // [B][P] SimMIM mask weights or null (model.py:28-33)
    const float *mask_tok = nullptr;  // [D]
};
hipError_t launch_patch_embed(int prec, const PatchArgs &pa, const void *w, const float *bias, const float *pos,
                              float *x, int dim, hipStream_t s, const StatsOut &so = StatsOut());

// ---- kernels_attn.hip
// head_dim: 64, or 128 in split-bf16 precision (q / k rows and V^T row groups are then 128 channels wide)
hipError_t launch_attention(int prec, const void *q, const void *k, const void *vt, void *ctx, float *lse2, int batch,
                            int n_tokens, int n_pad, int heads, float scale, hipStream_t s, int head_dim = 64,
                            float *ksplit_ws = nullptr, size_t ksplit_bytes = 0);
// bytes of key-slice workspace launch_attention can use at this size (0: it never splits the key range)
size_t attention_ksplit_bytes(int prec, int batch, int n_tokens, int heads, int head_dim);
hipError_t launch_attention_probs(int prec, const void *q, const void *k, const float *lse2, float *attn, int batch,
                                  int n_tokens, int n_pad, int heads, float scale, hipStream_t s, int head_dim = 64);
hipError_t launch_rows_from_probs(const float *attn, const int32_t *query_rows, int n_rows, float *rows, int batch,
                                  int n_tokens, int heads, hipStream_t s);
hipError_t launch_attention_rows(int prec, const void *q, const void *k, const int32_t *query_rows, int n_rows,
                                 float *rows, int batch, int n_tokens, int n_pad, int heads, float scale, hipStream_t s,
                                 int head_dim = 64);
hipError_t launch_attention_map(const float *attn, float *maps, int b, int heads, int n_tokens, int query, int hf,
                                int wf, int p, hipStream_t s);

// ---- kernels_post.hip (sliding-window post-processing)
hipError_t launch_tile_postprocess(const float *rows, float *maps, int tiles, int heads, int n_rows, int pixels,
                                   hipStream_t s);
hipError_t launch_bilinear_up(const float *src, float *dst, int tiles, int h, int w, int scale, hipStream_t s);
hipError_t launch_nearest_up(const float *src, float *dst, int tiles, int h, int w, int rep, hipStream_t s);
hipError_t launch_stitch(const float *crops, float *out, const double *ramp, int n, int window, int stride,
                         hipStream_t s);
hipError_t launch_normalize_u8(const float *img, size_t count, float *part, uint8_t *out,
                               unsigned long long *hist256, hipStream_t s);
hipError_t launch_threshold_u8(const uint8_t *img, uint8_t *mask, size_t count, int thresh, hipStream_t s);
hipError_t launch_tokens_to_fmap(const float *y, float *out, int batch, int n_tokens, int dim, hipStream_t s);
hipError_t launch_pixel_shuffle(const float *lin, float *out, int batch, int hp, int wp, int c_out, int sh, hipStream_t s);
hipError_t launch_head_mean(const float *rows, float *maps, int tiles, int heads, int n_rows, int pixels, hipStream_t s);
hipError_t launch_image_to_gray_u8(const float *img, int64_t stride_c, int chans, size_t count, uint8_t *out,
                                   unsigned long long *hist256, hipStream_t s);
hipError_t launch_blend_u8(const uint8_t *img, const uint8_t *att, size_t count, double alpha, double one_minus_alpha,
                           uint8_t *out, unsigned long long *hist256, hipStream_t s);

hipError_t launch_stitch_image_u8(const float *image, int64_t sc, int64_t sy, int chans, int H, int W, uint8_t *out,
                                  const double *ramp, int n, int window, int stride, unsigned long long *hist256,
                                  hipStream_t s);
hipError_t launch_weighted_u8(const float *heat, const uint8_t *img, size_t count, float *part, uint8_t *result,
                              uint8_t *att_u8, unsigned long long *hist_res, unsigned long long *hist_att, hipStream_t s);
hipError_t launch_histogram_u8(const uint8_t *img, size_t count, unsigned long long *hist256, hipStream_t s);
hipError_t launch_median_filter(const float *src, float *dst, int tiles, int h, int w, int k, hipStream_t s);
hipError_t launch_downscale_centre(const float *src, float *dst, int tiles, int h, int w, int f, hipStream_t s);

static inline int ocm_round_up(int v, int m) { return (v + m - 1) / m * m; }
// padded token count of the q / k / V^T buffers: the key axis of V^T is a contraction axis, so split pairs need whole
// groups of 32 keys
static inline int ocm_n_pad_for(int prec, int n) { return ocm_round_up(n, prec == 2 ? 32 : 8); }

// ---- Swin (kernels_swin.hip, kernels_gemm.hip)
hipError_t launch_linear_ld(int prec, const void *a, int64_t lda, const void *w, const float *bias, const float *resid,
                            void *out, int64_t ldo, int M, int N, int K, int epilogue, hipStream_t s);
hipError_t launch_swin_embed(const float *img, const float *w, const float *bias, const float *g, const float *be,
                             float *x, int batch, int chans, int size, int c0, float eps, hipStream_t s);
hipError_t launch_swin_ln(int prec, const float *x, const float *g, const float *be, void *y, size_t rows, int dim,
                          int ldy, float eps, bool merge, int Hin, int Win, hipStream_t s);
// SwinLayer.maybe_pad: operand rows to / the attention half's fp32 output from a grid padded to multiples of the window
hipError_t launch_swin_pad_rows(const void *src, void *dst, int batch, int H, int W, int Hp, int Wp, size_t row_bytes,
                                hipStream_t s);
hipError_t launch_swin_crop_add(float *x, const float *yp, int batch, int H, int W, int Hp, int Wp, int C, hipStream_t s);
hipError_t launch_swin_bias_perm(const float *table, float *perm, float *dense, int heads, int ws, hipStream_t s);
hipError_t launch_swin_window_attention(int prec, const void *qkv, int ld, void *ctx, int ldc, const float *bias_perm,
                                        const float *bias_dense, int batch, int H, int W, int ws, int shift, int heads,
                                        hipStream_t s);
// layernorm_before + q | k | v projection of a narrow-stage SwinLayer in one kernel (split-bf16, C = 96 / 128)
bool swin_lnqkv_fused_supported(int prec, int C);
hipError_t launch_swin_lnqkv(int prec, const float *x, const float *g, const float *be, const void *w, const float *bias,
                             void *qkv, size_t T, int C, float eps, hipStream_t s);
// the same kernel for any N (multiple of 32) output features, optionally with the exact-erf GELU: layernorm_after + mlp.fc1
hipError_t launch_swin_lnlinear(int prec, const float *x, const float *g, const float *be, const void *w, const float *bias,
                                void *out, size_t T, int C, int N, bool gelu, float eps, hipStream_t s);
// SwinLayer's MLP half in one kernel (split-bf16, C = 96 / 128, hidden = 4 C): x += fc2(gelu(fc1(LayerNorm(x))))
bool swin_mlp_fused_supported(int prec, int C, int hidden);
hipError_t launch_swin_mlp(int prec, float *x, const float *g, const float *be, const void *w1, const float *b1,
                           const void *w2, const float *b2, size_t T, int C, int hidden, float eps, hipStream_t s);
// SwinLayer's attention half in one kernel (split-bf16): C = 96 x += o_proj(window_attention(q | k | v of LayerNorm(x)));
// C = 192 ctx = window_attention(q | k | v of LayerNorm(x)) as pairs (swin_attn_block_proj_fused false: o_proj stays a GEMM)
bool swin_attn_block_fused_supported(int prec, int C, int heads, int ws);
bool swin_attn_block_proj_fused(int C);
hipError_t launch_swin_attn_block(int prec, float *x, const float *g, const float *be, const void *wqkv, const float *bqkv,
                                  const void *wo, const float *bo, const float *bias_perm, void *ctx, int batch, int H, int W,
                                  int ws, int shift, int heads, int C, float eps, hipStream_t s);
hipError_t launch_swin_pool_head(const float *x, const float *g, const float *be, const float *cw, const float *cb,
                                 float *logits, float *pooled, float *hidden, int batch, int L, int C, int labels,
                                 float eps, hipStream_t s);
hipError_t launch_cast_pad(int prec, const float *src, void *dst, size_t rows, int K, int Kp, hipStream_t s);
