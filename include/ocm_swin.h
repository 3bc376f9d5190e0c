/*
 * ocm_swin.h — C ABI of the Swin-T forward (BASELINE.json config 5), part of libocm_vit.so.
 *
 * Replaces, for inference, what Allen_data_Backbone/train.py:70-85 of linum-uqam/ViT-OCM-WMSegmentation
 * builds: transformers' SwinForImageClassification(SwinConfig(num_labels=5)) — patch embedding, four stages
 * of (shifted-)window attention blocks with relative-position bias, patch merging, final LayerNorm, mean pool
 * and classifier (transformers/models/swin/modeling_swin.py). Parameter names are that model's state_dict
 * keys. Same conventions as ocm_vit.h: plain pointers and sizes, int return codes (OCM_OK / OCM_E*),
 * ocm_last_error() for the message, nothing synchronises the device, no CPU fallback.
 */
#ifndef OCM_SWIN_H
#define OCM_SWIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ocm_swin_config {
    int32_t image_size;   /* square input, multiple of patch_size * window_size * 2^(stages-1) (224)   */
    int32_t patch_size;   /* 4                                                                        */
    int32_t num_channels; /* 1 or 3                                                                   */
    int32_t embed_dim;    /* C0 (96): stage s has C0 * 2^s channels                                   */
    int32_t num_stages;   /* 1..4                                                                     */
    int32_t depths[4];    /* (2, 2, 6, 2)                                                             */
    int32_t num_heads[4]; /* (3, 6, 12, 24): channels / heads must be 32                              */
    int32_t window_size;  /* 7                                                                        */
    int32_t num_labels;   /* classifier outputs (5 in the reference)                                  */
    float mlp_ratio;      /* 4.0                                                                      */
    float ln_eps;         /* 1e-5                                                                     */
    int32_t precision;    /* OCM_PREC_BF16 / OCM_PREC_FP32 / OCM_PREC_BF16X3 (ocm_vit.h)              */
    int32_t reserved;
} ocm_swin_config;

typedef struct ocm_swin ocm_swin_t;

int ocm_swin_create(const ocm_swin_config *cfg, ocm_swin_t **out);
void ocm_swin_destroy(ocm_swin_t *h);

/* Upload one parameter by its transformers state_dict key, e.g.
 *   swin.embeddings.patch_embeddings.projection.{weight,bias}, swin.embeddings.norm.{weight,bias},
 *   swin.encoder.layers.{s}.blocks.{b}.attention.{q,k,v,o}_proj.{weight,bias},
 *   swin.encoder.layers.{s}.blocks.{b}.attention.relative_position_bias.relative_position_bias_table,
 *   swin.encoder.layers.{s}.blocks.{b}.{layernorm_before,layernorm_after}.{weight,bias},
 *   swin.encoder.layers.{s}.blocks.{b}.mlp.{fc1,fc2}.{weight,bias},
 *   swin.encoder.layers.{s}.downsample.{reduction.weight,norm.weight,norm.bias},
 *   swin.layernorm.{weight,bias}, classifier.{weight,bias}
 * from a contiguous fp32 device buffer in the reference layout; the engine keeps its own packed copy. */
int ocm_swin_set_param(ocm_swin_t *h, const char *name, const float *dev_src, size_t count, void *stream);
int ocm_swin_params_ready(const ocm_swin_t *h);

size_t ocm_swin_workspace_bytes(const ocm_swin_t *h, int32_t batch);

/* SwinForImageClassification.forward (modeling_swin.py:1029-1066) on pixel_values (B, C, S, S) fp32, contiguous.
 * logits: (B, num_labels); pooled (optional): (B, C_last) = pooler_output; last_hidden (optional):
 * (B, L_last, C_last) = last_hidden_state after the final LayerNorm. */
int ocm_swin_forward(ocm_swin_t *h, const float *pixel_values, int32_t batch, float *logits, float *pooled,
                     float *last_hidden, void *workspace, size_t workspace_bytes, void *stream);

/* Per-handle dispatch options. OCM_SWIN_OPT_FUSE_MLP (default 1): in split-bf16 precision the layers of the narrow stages
 * (96 or 128 channels) run their MLP half as ONE kernel (ocm_op_swin_mlp), those of up to 192 channels layernorm_before +
 * the q | k | v projection as one (ocm_op_swin_lnqkv), and those of 96 channels their whole attention half as one
 * (ocm_op_swin_attn_block); 0 runs LayerNorm kernels, GEMMs and the window-attention kernel. Results agree to fp32 rounding. */
enum { OCM_SWIN_OPT_FUSE_MLP = 0 };
int ocm_swin_set_option(ocm_swin_t *h, int32_t option, int32_t value);

/* layernorm_before + the q | k | v projections of one SwinLayer (modeling_swin.py:641, :430-432) in one kernel:
 * qkv (tokens, 3 * channels) split pairs = LayerNorm(x; gamma, beta, eps) W^T + bias, W (3 * channels, channels) split pairs
 * (rows: q | k | v), x (tokens, channels) fp32. OCM_PREC_BF16X3, channels 96, 128 or 192; anything else returns OCM_EINVAL. */
int ocm_op_swin_lnqkv(int32_t precision, const float *x, const float *gamma, const float *beta, const void *w,
                      const float *bias, void *qkv, int64_t tokens, int32_t channels, float eps, void *stream);

/* MLP half of one SwinLayer (modeling_swin.py:668-672: layernorm_after, SwinIntermediate, SwinOutput, residual) in one
 * kernel: x (tokens, channels) fp32, in place, x += W2 gelu(W1 LayerNorm(x; gamma, beta, eps) + b1) + b2. Built for
 * precision OCM_PREC_BF16X3 (w1 (hidden, channels) and w2 (channels, hidden) as split pairs, ocm_op_cast_split),
 * channels 96 or 128, hidden = 4 x channels; anything else returns OCM_EINVAL. */
int ocm_op_swin_mlp(int32_t precision, float *x, const float *gamma, const float *beta, const void *w1, const float *b1,
                    const void *w2, const float *b2, int64_t tokens, int32_t channels, int32_t hidden, float eps,
                    void *stream);

/* Attention half of one SwinLayer (modeling_swin.py:641-666: layernorm_before, SwinSelfAttention with the relative-position
 * bias and the shift mask, SwinSelfOutput, residual) in one kernel: x (batch * height * width, 32 * heads) fp32, in place,
 * x += Wo window_attention(q | k | v of LayerNorm(x; gamma, beta, eps)) + bo. wqkv (3 * C, C) (rows q | k | v) and wo (C, C)
 * as split pairs (ocm_op_cast_split); rel_table: (2*ws-1)^2 x heads fp32 device table; scratch: heads * 4096 floats of device
 * memory, plus batch * height * width * C floats with 4 or 6 heads. Built for OCM_PREC_BF16X3 and 3 heads (C = 96, stage 0 of
 * Swin-T: one kernel), 4 heads (C = 128) or 6 heads (C = 192, stage 1 of Swin-T): one kernel up to the context + the o_proj GEMM;
 * anything else returns OCM_EINVAL. */
int ocm_op_swin_attn_block(int32_t precision, float *x, const float *gamma, const float *beta, const void *wqkv,
                           const float *bqkv, const void *wo, const float *bo, const float *rel_table, float *scratch,
                           int32_t batch, int32_t height, int32_t width, int32_t window, int32_t shift, int32_t heads, float eps,
                           void *stream);

/* Stand-alone (shifted-)window attention of one SwinLayer (modeling_swin.py:529-563 without the projections):
 * qkv (B*H*W, ld) holds q | k | v (heads*32 channels each) per token in E = bf16 / fp32 / split-bf16 pairs (`precision`; pairs: ld and ldc multiples of 32);
 * ctx (B*H*W, ldc) receives softmax(q k^T / sqrt(32) + bias + shift mask) v at the token's own row.
 * rel_table: (2*ws-1)^2 x heads fp32 device table; scratch: heads*(4096 + ws^4) floats of device memory. */
int ocm_op_swin_window_attention(int32_t precision, const void *qkv, int32_t ld, void *ctx, int32_t ldc,
                                 const float *rel_table, float *scratch, int32_t batch, int32_t height, int32_t width,
                                 int32_t window, int32_t shift, int32_t heads, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OCM_SWIN_H */
