/*
 * ocm_vit.h — C ABI of the MI355X (gfx950) ViT attention-map engine.
 *
 * The reference (linum-uqam/ViT-OCM-WMSegmentation) has NO native/FFI layer: its
 * boundary is the Python nn.Module surface of
 *   Self-supervised_segmentation/dino/vision_transformer.py
 * (SURVEY.md §8-b).  This header is therefore the ABI that the build's Python
 * mirror of that module binds through ctypes; every entry point names the
 * reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.
 *   - every pointer marked "dev" is a device (HBM) pointer owned by the caller
 *     unless stated otherwise; nothing here allocates caller-visible memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). No entry
 *     point synchronises the device; all work is enqueued on `stream`.
 *   - return value: 0 = OCM_OK, otherwise an OCM_E* code; ocm_last_error() gives
 *     the message of the last failure on the calling thread.
 *   - thread-compatible: one ocm_vit_t per model, no globals besides the
 *     thread-local error string.
 */
#ifndef OCM_VIT_H
#define OCM_VIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCM_ABI_VERSION 9

enum {
    OCM_OK = 0,
    OCM_EINVAL = 1,     /* bad argument / unsupported shape (Python: ValueError)      */
    OCM_ESTATE = 2,     /* parameter missing, handle not ready (Python: RuntimeError) */
    OCM_EHIP = 3,       /* HIP runtime / launch failure (Python: RuntimeError)        */
    OCM_ENOMEM = 4,     /* workspace too small / allocation failure                    */
    OCM_ENAME = 5       /* unknown parameter name (Python: KeyError)                   */
};

/* Arithmetic of the contraction kernels. The residual stream, LayerNorm
 * statistics, softmax and all accumulators are fp32 in every mode. */
enum {
    OCM_PREC_BF16 = 0, /* bf16 operands, v_mfma_f32_32x32x16_bf16, fp32 accumulate (fast path)            */
    OCM_PREC_FP32 = 1, /* fp32 operands, v_mfma_f32_32x32x2_f32: exact fp32 products, 1/16 the MFMA rate;
                          matrices, activations, q/k/v and P stay fp32 end to end (reference-grade maps) */
    OCM_PREC_BF16X3 = 2 /* split-bf16: every operand is a pair x = hi + lo of bf16 numbers (2^-17 relative) and every
                          product runs as three v_mfma_f32_32x32x16_bf16 (hi*hi + hi*lo + lo*hi), fp32 accumulate:
                          attention maps within 1e-3 of the fp32 reference on every golden weight set whose softmax is
                          not saturated — the trained-like sets of all three geometries, attention max 0.79 .. 0.96,
                          where single bf16 reaches 6e-2 — at ~3x the MFMA work of OCM_PREC_BF16 instead of 16x
                          (on a saturated softmax, attention max 1.0000, fp32 arithmetic itself is 5e-4 from
                          float64 and this mode 2-5e-3: DESIGN.md section 5). Split tensors ("E = split pairs" below) keep 4 bytes per element: the
                          contraction axis is cut into groups of 32 elements stored as 128 bytes
                          [32 x bf16 hi | 32 x bf16 lo] (ocm_op_cast_split / ocm_op_merge_split convert). */
};

/* Hyper-parameters: VisionTransformer.__init__ (vision_transformer.py:137-165) and
 * the vit_tiny/small/base factories (:259-279). head_dim = embed_dim/num_heads = 64 (T/S/B) runs the MFMA attention
 * kernels; any other multiple of 8 (the SimMIM encoder of model.py:93-103: 3 heads of 128) runs the fp32 attention
 * of kernels_attn.hip::attn_generic_kernel. */
typedef struct ocm_vit_config {
    int32_t patch_size;   /* p: 8 or 16 (any multiple of 8 up to 32)                   */
    int32_t in_chans;     /* 3 (reference) or 1 (grayscale-folded patch embedding)     */
    int32_t embed_dim;    /* D, multiple of 64                                         */
    int32_t depth;        /* L                                                         */
    int32_t num_heads;    /* H; head_dim = D/H                                         */
    int32_t mlp_hidden;   /* int(D*mlp_ratio), multiple of 64                          */
    float ln_eps;         /* 1e-6 for the DINO factories (:262,268,277)                */
    float qk_scale;       /* head_dim^-0.5 unless qk_scale was given (:71)             */
    int32_t precision;    /* OCM_PREC_*                                                */
    int32_t reserved;
} ocm_vit_config;

typedef struct ocm_vit ocm_vit_t; /* opaque engine handle */

/* ---- library --------------------------------------------------------------- */
int ocm_abi_version(void);
const char *ocm_last_error(void);

/* ---- handle life cycle ----------------------------------------------------- */
int ocm_vit_create(const ocm_vit_config *cfg, ocm_vit_t **out);
void ocm_vit_destroy(ocm_vit_t *h);

/* Upload one parameter. `name` is the reference state_dict key:
 *   cls_token, patch_embed.proj.{weight,bias}, blocks.{i}.norm1.{weight,bias},
 *   blocks.{i}.attn.qkv.{weight,bias}, blocks.{i}.attn.proj.{weight,bias},
 *   blocks.{i}.norm2.{weight,bias}, blocks.{i}.mlp.fc1.{weight,bias},
 *   blocks.{i}.mlp.fc2.{weight,bias}, norm.{weight,bias}
 * and, optionally, mask_token (model.py:16; only needed when ocm_vit_io.patch_mask is used)
 * (`pos_embed` is passed per forward, see ocm_vit_io.pos_embed).
 * `dev_src` is a contiguous fp32 device buffer of `count` elements in the
 * reference's layout; the engine keeps its own packed (bf16 for matrices, fp32
 * for vectors) copy, so the source may be freed after the stream has run. */
int ocm_vit_set_param(ocm_vit_t *h, const char *name, const float *dev_src, size_t count,
                      void *stream);
/* Per-handle dispatch options (this handle only; no process-wide state). 0 is always "automatic".
 *   OCM_OPT_FUSE_LN  attn.proj / mlp.fc2 + residual fused with the LayerNorm that follows (one full-row kernel, bit
 *                    identical to the GEMM + LayerNorm pair): 0 = where it is faster (>= 8192 token rows and fewer than
 *                    512 tiles of 128 x 128 in the output), 1 = never, 2 = whenever the embedding width has the kernel
 *                    (ocm_linear_resid_ln_supported).
 *   OCM_OPT_FOLD_LN  (OCM_PREC_BF16X3 engines) LayerNorm folded into the GEMM that consumes it: the residual stream is
 *                    handed over un-normalised as split pairs with per-row sums, the consumer multiplies by W * gamma
 *                    and finishes rstd * (acc - mu * c) + d in its epilogue — no LayerNorm launches, no full-row tiles.
 *                    0 = where it is faster (forwards whose (T x D) output holds fewer than 512 tiles of 128 x 128; the
 *                    block-level entry points keep the LayerNorm kernels), 1 = never, 2 = in every ocm_vit_forward. */
enum { OCM_OPT_FUSE_LN = 0, OCM_OPT_FOLD_LN = 1, OCM_OPT_COUNT = 2 };
int ocm_vit_set_option(ocm_vit_t *h, int32_t option, int32_t value);
/* 0 when every parameter has been set, else OCM_ESTATE with the first missing
 * name in ocm_last_error(). */
int ocm_vit_params_ready(const ocm_vit_t *h);

/* ---- forward --------------------------------------------------------------- */
enum {
    OCM_OUT_FEAT = 1 << 0,      /* norm(x) of the last n blocks       (get_intermediate_feat :234)  */
    OCM_OUT_ATTN = 1 << 1,      /* softmax probabilities (B,H,N,N)    (Attention.forward :83-85)    */
    OCM_OUT_QKV = 1 << 2,       /* (3,B,H,N,64) fp32                  (Attention.forward :80)       */
    OCM_OUT_TOKENS = 1 << 3,    /* raw residual stream after the last block (no final norm)         */
    OCM_OUT_ROWS = 1 << 4,      /* selected query rows of the last block's attention, CLS column
                                   dropped: (B,H,n_rows,N-1)          (utils.py:229-233)            */
    OCM_LAST_ATTN_ONLY = 1 << 5,/* get_last_selfattention (:239-246): the last block stops after
                                   its attention probabilities; FEAT/TOKENS/QKV must not be set.    */
    OCM_OUT_FMAP = 1 << 6,      /* norm(x)[:, 1:] as a (B,D,hp,wp) feature map — what the encoders of
                                   model.py:48-53,134-139 return                                   */
    OCM_USE_GRAPH = 1 << 7      /* launch through a cached hipGraph: the forward's ~80 kernel launches are
                                   captured once and replayed with one hipGraphLaunch while shapes, flags
                                   and every pointer in the io block stay the same (the steady state of the
                                   reference's one-tile-per-call loops); otherwise the graph is re-captured
                                   and updated in place. Same results, less host time per call.        */
};

/* One batch of tiles through prepare_tokens + blocks [+ final norm].
 * Tiles are p-aligned windows of fp32 planar images resident in HBM:
 *   pixel(b, c, y, x) = image[ b*img_stride_b + c*img_stride_c
 *                              + (y0_b + y)*img_stride_y + (x0_b + x) ]
 * where (y0_b, x0_b) = tile_origins[2b], tile_origins[2b+1] when tile_origins is
 * non-NULL and (0,0) otherwise.  A contiguous (B,3,H,W) torch tensor is
 * {stride_b = 3HW, stride_c = HW, stride_y = W}; a sliding-window sweep over one
 * slab uses stride_b = 0 and per-tile origins (sw_processing.py:151-163).
 * Strides are in elements.  x origins, img_stride_y and the base pointer must
 * keep every patch row 16-byte aligned (x0 % 4 == 0, stride_y % 4 == 0). */
typedef struct ocm_vit_io {
    const float *image;          /* dev */
    int64_t img_stride_b, img_stride_c, img_stride_y;
    const int32_t *tile_origins; /* dev, [batch][2] = (y0, x0), or NULL */
    int32_t batch;               /* B                                               */
    int32_t tile_h, tile_w;      /* pixels, multiples of patch_size                 */
    const float *pos_embed;      /* dev, [N][D] fp32: row 0 = cls position, rows 1.. = (interpolated)
                                    patch positions for this tile shape (:176-196)  */
    int32_t flags;               /* OCM_OUT_* | OCM_LAST_ATTN_ONLY                  */
    int32_t n_last;              /* n of get_intermediate_feat / get_intermediate_layers (>=1) */
    float *out_feat;             /* dev, [n_last][B][N][D]            or NULL       */
    float *out_attn;             /* dev, [n_last][B][H][N][N]         or NULL       */
    float *out_qkv;              /* dev, [n_last][3][B][H][N][hd]     or NULL       */
    float *out_tokens;           /* dev, [B][N][D]                    or NULL       */
    const int32_t *query_rows;   /* dev, [n_rows] token indices (0 = CLS) or NULL   */
    int32_t n_rows;
    int32_t reserved;
    float *out_rows;             /* dev, [B][H][n_rows][N-1]          or NULL       */
    const float *patch_mask;     /* dev, [B][P] fp32 SimMIM mask w or NULL: patch rows become
                                    x*(1-w) + mask_token*w before cls/pos (model.py:28-33); needs the
                                    optional parameter "mask_token"                 */
    float *out_fmap;             /* dev, [B][D][hp][wp]               or NULL       */
    void *workspace;             /* dev, >= ocm_vit_workspace_bytes(h, batch, N)    */
    size_t workspace_bytes;
    void *stream;                /* hipStream_t                                     */
} ocm_vit_io;

/* Bytes of scratch HBM one forward of `batch` tiles with `n_tokens` = N needs. */
size_t ocm_vit_workspace_bytes(const ocm_vit_t *h, int32_t batch, int32_t n_tokens);

/* VisionTransformer.get_intermediate_feat / get_last_selfattention / forward_feats /
 * get_intermediate_layers (vision_transformer.py:211-256), selected by io->flags. */
int ocm_vit_forward(ocm_vit_t *h, const ocm_vit_io *io);

/* Counters of the OCM_USE_GRAPH path: forwards replayed from the cached graph / graph (re-)captures. */
int ocm_vit_graph_stats(const ocm_vit_t *h, uint64_t *replays, uint64_t *captures);

/* VisionTransformer.prepare_tokens (:198-209): patch embedding + cls + pos.
 * Uses image/strides/origins/batch/tile_*, pos_embed and stream of `io`;
 * writes x_out[B][N][D] fp32. */
int ocm_vit_prepare_tokens(ocm_vit_t *h, const ocm_vit_io *io, float *x_out);

/* Block.forward (:106-114) of block `index`, in place on x[B][N][D] fp32.
 * flags: OCM_OUT_ATTN / OCM_OUT_QKV fill out_attn (B,H,N,N) / out_qkv (3,B,H,N,64);
 * OCM_LAST_ATTN_ONLY returns after the probabilities (return_attention=True). */
int ocm_vit_block_forward(ocm_vit_t *h, int32_t index, float *x, int32_t batch, int32_t n_tokens,
                          int32_t flags, float *out_attn, float *out_qkv, void *workspace,
                          size_t workspace_bytes, void *stream);

/* self.norm (:158, :214, :221, :234): y = LayerNorm(x) over rows of D, fp32 out. */
int ocm_vit_final_norm(ocm_vit_t *h, const float *x, float *y, int64_t rows, void *stream);

/* ---- stand-alone operators (kernel-level parity tests; same kernels the engine uses) ---- */

/* nn.LayerNorm(D, eps) (:98,102,158). y is fp32, bf16 or split pairs (dim % 32 == 0) by `out_kind`. */
enum { OCM_LN_F32 = 0, OCM_LN_BF16 = 1, OCM_LN_SPLIT = 2 };
int ocm_op_layernorm(const float *x, const float *gamma, const float *beta, void *y,
                     int32_t out_kind, int64_t rows, int32_t dim, float eps, void *stream);

/* fp32 -> bf16 (round-to-nearest-even) of `count` elements. */
int ocm_op_cast_bf16(const float *src, void *dst_bf16, size_t count, void *stream);
/* fp32 -> split pairs and back (hi + lo in fp32); `count` % 32 == 0, dst of 4*count bytes. Rows whose length is a
 * multiple of 32 keep their row structure (row r starts at byte r * 4 * row_len). */
int ocm_op_cast_split(const float *src, void *dst_split, size_t count, void *stream);
int ocm_op_merge_split(const void *src_split, float *dst, size_t count, void *stream);

enum {
    OCM_EPI_BIAS_F32 = 0,       /* out fp32 [M][N] = acc + bias                                  */
    OCM_EPI_BIAS_RESID_F32 = 1, /* out fp32 [M][N] = resid + acc + bias   (proj/fc2 + skip :110) */
    OCM_EPI_BIAS_GELU_BF16 = 2, /* out bf16 [M][N] = gelu_erf(acc + bias) (fc1 + act :58-59)     */
    OCM_EPI_BIAS_BF16 = 3       /* out bf16 [M][N] = acc + bias                                  */
};
/* The stand-alone contraction operators take `precision` (OCM_PREC_*): operand / activation buffers
 * named E are bf16 for OCM_PREC_BF16, fp32 for OCM_PREC_FP32 and split pairs for OCM_PREC_BF16X3 (the *_BF16
 * epilogues then emit E). */

/* nn.Linear: out = epilogue(A[M][K] · W[N][K]^T + bias[N]); A, W of type E row-major,
 * K % 64 == 0 (bf16) or K % 32 == 0 (fp32, split pairs), N % 32 == 0. resid may alias out. */
int ocm_op_linear(int32_t precision, const void *a, const void *w, const float *bias, const float *resid,
                  void *out, int32_t M, int32_t N, int32_t K, int32_t epilogue, void *stream);

/* nn.Linear + residual + LayerNorm in one kernel (Block.forward :107-111 with the NEXT normalisation folded in:
 *   x = resid + A[M][K] · W[D][K]^T + bias;   xn = LayerNorm(x; gamma, beta, eps) in the activation type E
 * exactly as ocm_op_linear (epilogue RESID_F32) followed by ocm_op_layernorm would produce them (bit for bit).
 * A workgroup owns whole rows, so D is one of 128 / 256 / 384 (ocm_linear_resid_ln_supported). resid may
 * alias x. The engine uses it for attn.proj -> norm2 and mlp.fc2 -> the next block's norm1 when M >= 8192. */
int ocm_linear_resid_ln_supported(int32_t D);
int ocm_op_linear_resid_ln(int32_t precision, const void *a, const void *w, const float *bias, const float *resid,
                           float *x, const float *gamma, const float *beta, void *xn, int32_t M, int32_t D, int32_t K,
                           float eps, void *stream);

/* Head-major packed projections the attention kernels consume:
 *   q, k : E [B*H][n_pad][64];  vt : E [B*H][64][n_pad],  n_pad = ocm_n_pad_prec(precision, N)
 * (N rounded up to 8, or to 32 for split pairs: the key axis of V^T is a contraction axis). ocm_n_pad(N) is the
 * bf16 / fp32 value. */
int32_t ocm_n_pad(int32_t n_tokens);
int32_t ocm_n_pad_prec(int32_t precision, int32_t n_tokens);

/* Attention.forward qkv projection (:80): A[B*N][D] bf16 · Wqkv[3D][D]^T + b -> q/k/vt
 * (and, when qkv_f32 != NULL, the fp32 (3,B,H,N,64) tensor the reference returns). */
int ocm_op_qkv_proj(int32_t precision, const void *a, const void *w, const float *bias, void *q, void *k,
                    void *vt, float *qkv_f32, int32_t batch, int32_t n_tokens, int32_t heads,
                    void *stream);

/* softmax(q k^T * scale) v (:83-87). ctx E [B][N][H*64] (or NULL);
 * lse2 fp32 [B*H][N] = log2-domain log-sum-exp of the scaled scores (or NULL). */
int ocm_op_attention(int32_t precision, const void *q, const void *k, const void *vt, void *ctx, float *lse2,
                     int32_t batch, int32_t n_tokens, int32_t heads, float scale, void *stream);

/* Attention probabilities (:83-84) from q, k and lse2: attn fp32 [B][H][N][N]. */
int ocm_op_attention_probs(int32_t precision, const void *q, const void *k, const float *lse2, float *attn,
                           int32_t batch, int32_t n_tokens, int32_t heads, float scale,
                           void *stream);

/* Selected rows of the probabilities with the CLS column dropped
 * (utils.py:232, attentions[0, :, query, 1:]): rows fp32 [B][H][n_rows][N-1]. */
int ocm_op_attention_rows(int32_t precision, const void *q, const void *k, const int32_t *query_rows,
                          int32_t n_rows, float *rows, int32_t batch, int32_t n_tokens, int32_t heads, float scale,
                          void *stream);

/* The qkv projection, attention and probabilities for heads of `head_dim` channels: 64 (identical to the entry points
 * above) or 128 — the encoder the reference's build_model() constructs (Self-supervised_segmentation/model.py:93-103:
 * embed_dim 384, 3 heads) — in every precision (split-bf16 since round 3, fp32 and single bf16 since round 4). q / k are then
 * [B*H][n_pad][128] and vt [B*H][128][n_pad] in the operand type, ctx [B][N][H*128]. Any other width returns OCM_EINVAL (an
 * engine handle runs such heads on its generic fp32 attention kernel) — except that ocm_op_qkv_proj_hd with q = k = vt = NULL
 * fills only qkv_f32, for any head_dim that is a multiple of 8 (the input of ocm_op_attention_generic). */
int ocm_op_qkv_proj_hd(int32_t precision, const void *a, const void *w, const float *bias, void *q, void *k, void *vt,
                       float *qkv_f32, int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, void *stream);
int ocm_op_attention_hd(int32_t precision, const void *q, const void *k, const void *vt, void *ctx, float *lse2,
                        int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, float scale, void *stream);
int ocm_op_attention_probs_hd(int32_t precision, const void *q, const void *k, const float *lse2, float *attn,
                              int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, float scale, void *stream);

/* Attention for heads of ANY width (head_dim a multiple of 4, up to 512; at most 8192 tokens) on the fp32 tensor
 * qkv_f32 [3][B][H][N][head_dim] that ocm_op_qkv_proj_hd writes: plain fp32 FMAs, one wavefront per query row — what an engine
 * handle runs for head widths the MFMA kernels are not built for, and what the free-standing Attention module
 * (dino/vision_transformer.py:66-90 accepts any dim / num_heads) calls for them. ctx (optional): [B][N][H*head_dim] in the
 * operand type of `precision`; attn (optional): fp32 [B][H][N][N]. */
int ocm_op_attention_generic(int32_t precision, const float *qkv_f32, void *ctx, float *attn, int32_t batch,
                             int32_t n_tokens, int32_t heads, int32_t head_dim, float scale, void *stream);

/* compute_attention (utils.py:229-235) on device: attn fp32 [B][H][N][N] ->
 * maps fp32 [H][hf*p][wf*p] = nearest-neighbour x p upsample of attn[b, :, query, 1:]. */
int ocm_op_attention_map(const float *attn, float *maps, int32_t b, int32_t heads,
                         int32_t n_tokens, int32_t query, int32_t hf, int32_t wf, int32_t p,
                         void *stream);

/* ---- in-library kernel timing (diagnostic; process-wide, single host thread) --------------------
 * While enabled, every launch of a kernel class selected in `class_mask` is bracketed by two
 * hipEvents recorded on the launch stream (no host synchronisation). ocm_prof_end() waits for the
 * recorded events, returns per-class launch counts and summed device milliseconds, and disables
 * profiling. bench.py uses it for the live per-kernel duration behind its `roofline` object. */
enum {
    OCM_K_PATCH = 0, /* patch-embedding gather + GEMM                 */
    OCM_K_LN = 1,    /* LayerNorm                                      */
    OCM_K_QKV = 2,   /* qkv projection GEMM                            */
    OCM_K_ATTN = 3,  /* flash attention (scores + softmax + P.V)       */
    OCM_K_PROBS = 4, /* attention-probability materialisation / rows   */
    OCM_K_PROJ = 5,  /* attn.proj GEMM + residual                      */
    OCM_K_FC1 = 6,   /* mlp.fc1 GEMM + GELU                            */
    OCM_K_FC2 = 7,   /* mlp.fc2 GEMM + residual                        */
    OCM_K_COUNT = 8
};
int ocm_prof_begin(uint32_t class_mask, int32_t max_launches);
int ocm_prof_end(double *ms_per_class /*[OCM_K_COUNT]*/, int64_t *launches_per_class /*[OCM_K_COUNT]*/);

/* ---- sliding-window post-processing on device (SURVEY §8-f "next" rows 1-2; sw_processing.py) ---- */
/* :245 + :253-254 — rows (T,H,n_rows,P) CLS-row maps (row 0 used) -> maps (T,P):
 * head mean (sequential fp32, as np.mean(axis=0)) then per-window (v - min) / (max - min) * 255. */
int ocm_op_tile_postprocess(const float *rows, float *maps, int32_t tiles, int32_t heads, int32_t n_rows,
                            int32_t pixels, void *stream);
/* :255-257 — cv2.resize(INTER_LINEAR) x`scale` of (T,h,w) float32 maps: half-pixel centres, replicate
 * border, fp32 (cv2 is an un-vendored dependency: parity unpinned). */
int ocm_op_bilinear_upsample(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t scale,
                             void *stream);
/* Nearest-neighbour x`rep` of (T,h,w) float32 maps (index replication: compute_attention's nearest upsample utils.py:233,
 * the block values of the //8 *8 resize chain sw_processing.py:255-257, the patch mask of model.py:71): dst (T,h*rep,w*rep). */
int ocm_op_nearest_upsample(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t rep, void *stream);
/* concat_crops :113-149 — n x n row-major float32 windows (n*n, window, window) -> (S,S),
 * S = window + (n-1)*stride; ramp = np.linspace(1, 0, window - stride) as float64 on device.
 * Bit-exact with the reference's sequential stitcher. stride < window <= 3*stride. */
int ocm_op_stitch(const float *crops, float *out, const double *ramp, int32_t n, int32_t window,
                  int32_t stride, void *stream);
/* threshold() :43-48 — min_max_normalize -> *255 -> astype(uint8) and the 256-bin histogram of the result.
 * scratch: >= 2048 bytes of device memory. */
int ocm_op_normalize_u8(const float *img, int64_t count, void *scratch, uint8_t *out, uint64_t *hist256,
                        void *stream);
/* cv2.threshold(img, 0, 255, THRESH_BINARY + THRESH_OTSU) :62 — host part: Otsu level of a histogram
 * (restates OpenCV's getThreshVal_Otsu_8u; parity unpinned), and the binary mask on device. */
int32_t ocm_otsu_threshold(const uint64_t *hist256_host, int64_t count);
int ocm_op_threshold_u8(const uint8_t *img, uint8_t *mask, int64_t count, int32_t thresh, void *stream);

/* sw_processing.py:224-227 — the stitched image of the windows, as the reference builds it for threshold():
 * concat_crops on the uint8 RGB windows (float64 blend, TRUNCATED into the uint8 overlap array at every fold), then
 * PIL .convert("L"). `image`: the float slab planes (x = u / 255 as ToTensor made them; 1 or 3 planes, strides in
 * elements), height x width pixels; windows that reach past the slab are zero-filled (PIL crop). out: (S,S) uint8,
 * S = window + (n-1)*stride; hist256 optional. */
int ocm_op_stitch_image_u8(const float *image, int64_t stride_c, int64_t stride_y, int32_t chans, int32_t height,
                           int32_t width, uint8_t *out, const double *ramp, int32_t n, int32_t window, int32_t stride,
                           uint64_t *hist256, void *stream);
/* sw_processing.py:42-48 — attention = min_max_normalize(heat); result = (img * attention / max(attention)).astype(uint8)
 * (float32 arithmetic, truncation) and att_u8 = (attention * 255).astype(uint8), with both 256-bin histograms.
 * scratch: >= 2048 bytes. */
int ocm_op_weighted_u8(const float *heat, const uint8_t *img, int64_t count, void *scratch, uint8_t *result,
                       uint8_t *att_u8, uint64_t *hist_result, uint64_t *hist_att, void *stream);

/* 256-bin histogram of a uint8 image (input of the Otsu levels when the image arrives as uint8). */
int ocm_op_histogram_u8(const uint8_t *img, int64_t count, uint64_t *hist256, void *stream);

/* eval.py:144,158 — scipy.ndimage.median_filter(map, size): size x size footprint, mode "reflect", upper median.
 * (T,h,w) fp32 -> (T,h,w); src != dst. Pinned against scipy (tests/golden/median.npz). */
int ocm_op_median_filter(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t size, void *stream);
/* eval.py:169, sw_processing.py:255 — cv2.resize(map, (w/f, h/f)) (INTER_LINEAR) for an integer factor: the two
 * centre pixels averaged per axis in float32 (cv2: parity unpinned). (T,h,w) -> (T,h/f,w/f). */
int ocm_op_downscale_centre(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t factor,
                            void *stream);

/* ---- eval.py's per-image mask chain (eval.py:126-171, utils.py:55-115 threshold()) ---- */
/* eval.py:142 — np.mean(attention_response, axis=0): rows (T,H,n_rows,P) -> maps (T,P), sequential fp32. */
int ocm_op_head_mean(const float *rows, float *maps, int32_t tiles, int32_t heads, int32_t n_rows,
                     int32_t pixels, void *stream);
/* eval.py:122,166 — transform(img).convert("L"): float planes (1 or 3, `stride_c` elements apart) ->
 * uint8 by mul(255).byte() and PIL's integer RGB->L weights; optional 256-bin histogram. */
int ocm_op_image_to_gray_u8(const float *image, int64_t stride_c, int32_t chans, int64_t count, uint8_t *out,
                            uint64_t *hist256, void *stream);
/* utils.py:76-81 — result = ((img / 2) * (1 - alpha) + (attention / 2) * alpha).astype(uint8) in float64;
 * `one_minus_alpha` is the host's own (1 - alpha) double. Optional histogram of the result. */
int ocm_op_blend_u8(const uint8_t *img, const uint8_t *att, int64_t count, double alpha, double one_minus_alpha,
                    uint8_t *out, uint64_t *hist256, void *stream);

/* ---- decoder head of model.py:60-66,147-152: Conv2d(D, s*s*C_out, 1) + PixelShuffle(s) ---- */
/* lin: (B*hp*wp, s*s*c_out) fp32 = the 1x1 conv evaluated token-major (ocm_op_linear on the normed patch
 * tokens); out: (B, c_out, hp*s, wp*s) with out[b][c][y*s+i][x*s+j] = lin[b*hp*wp + y*wp + x][c*s*s + i*s + j]. */
int ocm_op_pixel_shuffle(const float *lin, float *out, int32_t batch, int32_t hp, int32_t wp, int32_t c_out,
                         int32_t s, void *stream);

/* ---- two-layer decoder of model.py:154-166: Conv2d(3x3, padding 1) as im2col + ocm_op_linear ---- */
/* in: token-major fp32 [B][h*w][C] (C % 32 == 0); out: operand rows E [B*h*w][9*C], K index (ky*3 + kx)*C + c, zeros
 * outside the image; relu != 0 applies max(x, 0) on the way (the nn.ReLU in front of the second convolution). The
 * matching weight is the (O, C, 3, 3) kernel permuted to (O, 3, 3, C). */
int ocm_op_im2col3x3(int32_t precision, const float *in, void *out, int32_t batch, int32_t h, int32_t w,
                     int32_t channels, int32_t relu, void *stream);

/* ---- sliding-window index math (host, integer; sw_processing.py:151-163) ---- */
/* Number of windows per axis: len(range(0, size - 2*stride, stride)). */
int32_t ocm_sw_count(int32_t size, int32_t stride);
/* Row-major (y0, x0) origins for an (height x width) image; returns the number of
 * windows written (<= cap) or a negative OCM_E* code. */
int32_t ocm_sw_origins(int32_t height, int32_t width, int32_t stride, int32_t *origins_yx,
                       int32_t cap);
/* Contiguous block partition of `n_tiles` over `world` ranks:
 * rank r owns [*begin, *end); every rank's padded share is ceil(n_tiles/world). */
int32_t ocm_sw_shard(int32_t n_tiles, int32_t world, int32_t rank, int32_t *begin, int32_t *end);

#ifdef __cplusplus
}
#endif
#endif /* OCM_VIT_H */
