// launch.h — host-side launchers shared between the kernel translation units and the
// engine (engine.hip). Every launcher only enqueues work on `s`; none synchronises.
#pragma once
#include "common.h"

// ---- kernels_misc.hip
hipError_t launch_layernorm(const float *x, const float *gamma, const float *beta, void *y, bool out_bf16,
                            int64_t rows, int dim, float eps, hipStream_t s);
hipError_t launch_cast_bf16(const float *src, bf16 *dst, size_t count, hipStream_t s);
// sum over the channel axis of a (D, C, p, p) conv weight -> bf16 (D, p*p): grayscale fold.
hipError_t launch_fold_cast_bf16(const float *src, bf16 *dst, int D, int C, int pp, hipStream_t s);
hipError_t launch_cls_rows(const float *cls, const float *pos, float *x, int batch, int n_tokens, int dim,
                           hipStream_t s);

// ---- kernels_gemm.hip
hipError_t launch_linear(const bf16 *a, const bf16 *w, const float *bias, const float *resid, void *out, int M,
                         int N, int K, int epilogue, hipStream_t s);
hipError_t launch_qkv(const bf16 *a, const bf16 *w, const float *bias, bf16 *q, bf16 *k, bf16 *vt,
                      float *qkv_f32, int batch, int n_tokens, int n_pad, int heads, hipStream_t s);
struct PatchArgs {
    const float *image;
    int64_t sb, sc, sy;
    const int32_t *origins;
    int batch, hp, wp, p, chans;  // hp x wp patches per tile
};
hipError_t launch_patch_embed(const PatchArgs &pa, const bf16 *w, const float *bias, const float *pos, float *x,
                              int dim, hipStream_t s);

// ---- kernels_attn.hip
hipError_t launch_attention(const bf16 *q, const bf16 *k, const bf16 *vt, bf16 *ctx, float *lse2, int batch,
                            int n_tokens, int n_pad, int heads, float scale, hipStream_t s);
hipError_t launch_attention_probs(const bf16 *q, const bf16 *k, const float *lse2, float *attn, int batch,
                                  int n_tokens, int n_pad, int heads, float scale, hipStream_t s);
hipError_t launch_attention_rows(const bf16 *q, const bf16 *k, const int32_t *query_rows, int n_rows, float *rows,
                                 int batch, int n_tokens, int n_pad, int heads, float scale, hipStream_t s);
hipError_t launch_attention_map(const float *attn, float *maps, int b, int heads, int n_tokens, int query, int hf,
                                int wf, int p, hipStream_t s);

static inline int ocm_round_up(int v, int m) { return (v + m - 1) / m * m; }
