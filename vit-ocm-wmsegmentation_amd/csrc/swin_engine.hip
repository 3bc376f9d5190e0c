// swin_engine.hip — the C ABI of include/ocm_swin.h: parameter store and launch sequence of one Swin forward
// (transformers/models/swin/modeling_swin.py, the model Allen_data_Backbone/train.py:70-85 of the reference
// builds). Host code only; kernels live in kernels_swin.hip / kernels_gemm.hip. Nothing here synchronises.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/ocm_swin.h"
#include "host_common.h"
#include "launch.h"

#define fail ocm_fail

namespace {

enum SlotKind {
    S_F32,    // fp32 vector / matrix copied as is
    S_MAT,    // [rows][K] -> E [rows][Kp] (zero padded), optionally at a row offset of a fused matrix
    S_TABLE,  // relative-position bias table -> permuted (+ dense) bias
};

struct Slot {
    std::string name;
    SlotKind kind;
    size_t count;        // elements of the reference tensor
    size_t offset;       // bytes into the arena (destination base)
    int rows, K, Kp;     // S_MAT geometry
    size_t row_offset;   // S_MAT: first destination row (fused q|k|v)
    int heads;           // S_TABLE
    size_t dense_offset; // S_TABLE: dense [heads][A][A] table (fp32 kernel)
    bool set;
};

struct LayerP {
    size_t ln1_g, ln1_b, ln2_g, ln2_b;  // arena offsets
    size_t wqkv, bqkv, wo, bo, w1, b1, w2, b2, bias_perm, bias_dense;  // arena offsets
};

struct StageP {
    std::vector<LayerP> layers;
    size_t red_w, red_g, red_b;  // patch merging (unused for the last stage)
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

struct ocm_swin {
    ocm_swin_config cfg;
    int prec;
    size_t esz;
    int kstep;  // elements per GEMM K step: 64 (bf16) / 32 (fp32)
    std::vector<Slot> slots;
    std::vector<StageP> stages;
    size_t emb_w, emb_b, emb_g, emb_be, fin_g, fin_b, cls_w, cls_b;
    char *arena = nullptr;
    size_t arena_bytes = 0;
    bool fuse_mlp = true;  // ocm_swin_set_option(OCM_SWIN_OPT_FUSE_MLP)

    size_t reserve(size_t bytes) {
        const size_t off = arena_bytes;
        arena_bytes += (bytes + 255) & ~(size_t)255;
        return off;
    }
    int add(const std::string &name, SlotKind kind, size_t count, size_t offset, int rows = 0, int K = 0, int Kp = 0,
            size_t row_offset = 0, int heads = 0, size_t dense_offset = 0) {
        slots.push_back(Slot{name, kind, count, offset, rows, K, Kp, row_offset, heads, dense_offset, false});
        return (int)slots.size() - 1;
    }
    size_t add_f32(const std::string &name, size_t count) {
        const size_t off = reserve(count * 4);
        add(name, S_F32, count, off);
        return off;
    }
    template <class T>
    T *ptr(size_t off) const {
        return (T *)(arena + off);
    }
    int Kp(int K) const { return round_up(K, kstep); }
    int chans(int s) const { return cfg.embed_dim << s; }
};

extern "C" int ocm_swin_create(const ocm_swin_config *cfg, ocm_swin_t **out) {
    if (!cfg || !out) return fail(OCM_EINVAL, "ocm_swin_create: null argument");
    if (cfg->patch_size != 4) return fail(OCM_EINVAL, "patch_size %d: only 4 is built", cfg->patch_size);
    if (cfg->num_channels != 1 && cfg->num_channels != 3) return fail(OCM_EINVAL, "num_channels must be 1 or 3");
    if (cfg->window_size < 2 || cfg->window_size > 7)
        return fail(OCM_EINVAL, "window_size %d must be in [2,7] (one 64-position tile per window)", cfg->window_size);
    if (cfg->num_stages < 1 || cfg->num_stages > 4) return fail(OCM_EINVAL, "num_stages %d must be in [1,4]", cfg->num_stages);
    if (cfg->embed_dim <= 0 || cfg->embed_dim % 32 || cfg->embed_dim > 128)
        return fail(OCM_EINVAL, "embed_dim %d must be a multiple of 32, <= 128", cfg->embed_dim);
    if (cfg->num_labels <= 0) return fail(OCM_EINVAL, "num_labels must be positive");
    if (cfg->precision != OCM_PREC_BF16 && cfg->precision != OCM_PREC_FP32 && cfg->precision != OCM_PREC_BF16X3)
        return fail(OCM_EINVAL, "bad precision");
    int grid = cfg->image_size / 4;
    if (cfg->image_size <= 0 || cfg->image_size % 4) return fail(OCM_EINVAL, "image_size must be a multiple of 4");
    for (int s = 0; s < cfg->num_stages; ++s) {
        const int C = cfg->embed_dim << s;
        if (cfg->depths[s] <= 0) return fail(OCM_EINVAL, "depths[%d] must be positive", s);
        if (cfg->num_heads[s] <= 0 || C != cfg->num_heads[s] * 32)
            return fail(OCM_EINVAL, "stage %d: head_dim must be 32 (channels %d, heads %d)", s, C, cfg->num_heads[s]);
        // a grid that is not a multiple of the window is padded (SwinLayer.maybe_pad), an odd one gets a row / column of zeros in
        // the patch merging (SwinPatchMerging.maybe_pad); a grid SMALLER than the window is what transformers itself cannot
        // run (set_shift_and_window_size shrinks the window, the relative-position bias keeps the configured size)
        if (grid < cfg->window_size)
            return fail(OCM_EINVAL, "stage %d grid %d is smaller than window_size %d", s, grid, cfg->window_size);
        if (s + 1 < cfg->num_stages) grid = (grid + 1) / 2;
    }
    const int M4 = (int)(cfg->mlp_ratio * cfg->embed_dim);
    if (M4 <= 0 || M4 % 32) return fail(OCM_EINVAL, "mlp_ratio * embed_dim must be a multiple of 32");

    ocm_swin *h = new ocm_swin();
    h->cfg = *cfg;
    h->prec = cfg->precision == OCM_PREC_FP32 ? 1 : cfg->precision == OCM_PREC_BF16X3 ? 2 : 0;
    h->esz = h->prec ? 4 : 2;       // split-bf16 pairs are 4 bytes per element too
    h->kstep = h->prec ? 32 : 64;   // and their K step is one 128-byte group of 32
    const int ws = cfg->window_size, A = ws * ws;
    const std::string e = "swin.embeddings.";
    const int C0 = cfg->embed_dim, Kpe = cfg->num_channels * 16;
    h->emb_w = h->add_f32(e + "patch_embeddings.projection.weight", (size_t)C0 * Kpe);
    h->emb_b = h->add_f32(e + "patch_embeddings.projection.bias", C0);
    h->emb_g = h->add_f32(e + "norm.weight", C0);
    h->emb_be = h->add_f32(e + "norm.bias", C0);
    h->stages.resize(cfg->num_stages);
    for (int s = 0; s < cfg->num_stages; ++s) {
        const int C = h->chans(s), heads = cfg->num_heads[s], M = (int)(cfg->mlp_ratio * C);
        const int Kc = h->Kp(C), Km = h->Kp(M);
        StageP &st = h->stages[s];
        for (int b = 0; b < cfg->depths[s]; ++b) {
            const std::string pre = "swin.encoder.layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
            LayerP lp{};
            lp.wqkv = h->reserve((size_t)3 * C * Kc * h->esz);
            lp.bqkv = h->reserve((size_t)3 * C * 4);
            const char *nm[3] = {"q_proj", "k_proj", "v_proj"};
            for (int i = 0; i < 3; ++i) {
                h->add(pre + "attention." + nm[i] + ".weight", S_MAT, (size_t)C * C, lp.wqkv, C, C, Kc, (size_t)i * C);
                h->add(pre + "attention." + nm[i] + ".bias", S_F32, C, lp.bqkv + (size_t)i * C * 4);
            }
            lp.wo = h->reserve((size_t)C * Kc * h->esz);
            h->add(pre + "attention.o_proj.weight", S_MAT, (size_t)C * C, lp.wo, C, C, Kc);
            lp.bo = h->add_f32(pre + "attention.o_proj.bias", C);
            lp.bias_perm = h->reserve((size_t)heads * 2 * 64 * 32 * 4);
            lp.bias_dense = h->reserve((size_t)heads * A * A * 4);
            h->add(pre + "attention.relative_position_bias.relative_position_bias_table", S_TABLE,
                   (size_t)(2 * ws - 1) * (2 * ws - 1) * heads, lp.bias_perm, 0, 0, 0, 0, heads, lp.bias_dense);
            const size_t g1 = h->add_f32(pre + "layernorm_before.weight", C), b1 = h->add_f32(pre + "layernorm_before.bias", C);
            const size_t g2 = h->add_f32(pre + "layernorm_after.weight", C), b2 = h->add_f32(pre + "layernorm_after.bias", C);
            lp.ln1_g = g1;
            lp.ln1_b = b1;
            lp.ln2_g = g2;
            lp.ln2_b = b2;
            lp.w1 = h->reserve((size_t)M * Kc * h->esz);
            h->add(pre + "mlp.fc1.weight", S_MAT, (size_t)M * C, lp.w1, M, C, Kc);
            lp.b1 = h->add_f32(pre + "mlp.fc1.bias", M);
            lp.w2 = h->reserve((size_t)C * Km * h->esz);
            h->add(pre + "mlp.fc2.weight", S_MAT, (size_t)C * M, lp.w2, C, M, Km);
            lp.b2 = h->add_f32(pre + "mlp.fc2.bias", C);
            st.layers.push_back(lp);
        }
        if (s + 1 < cfg->num_stages) {
            const std::string pre = "swin.encoder.layers." + std::to_string(s) + ".downsample.";
            const int K4 = h->Kp(4 * C);
            st.red_w = h->reserve((size_t)2 * C * K4 * h->esz);
            h->add(pre + "reduction.weight", S_MAT, (size_t)2 * C * 4 * C, st.red_w, 2 * C, 4 * C, K4);
            st.red_g = h->add_f32(pre + "norm.weight", 4 * C);
            st.red_b = h->add_f32(pre + "norm.bias", 4 * C);
        }
    }
    const int Cl = h->chans(cfg->num_stages - 1);
    h->fin_g = h->add_f32("swin.layernorm.weight", Cl);
    h->fin_b = h->add_f32("swin.layernorm.bias", Cl);
    h->cls_w = h->add_f32("classifier.weight", (size_t)cfg->num_labels * Cl);
    h->cls_b = h->add_f32("classifier.bias", cfg->num_labels);
    if (h->arena_bytes > ((size_t)1 << 31)) {
        delete h;
        return fail(OCM_EINVAL, "parameter arena too large");
    }
    hipError_t er = hipMalloc((void **)&h->arena, h->arena_bytes);
    if (er != hipSuccess) {
        const size_t bytes = h->arena_bytes;
        delete h;
        return fail(OCM_ENOMEM, "hipMalloc(%zu) for Swin parameters failed: %s", bytes, hipGetErrorString(er));
    }
    er = hipMemset(h->arena, 0, h->arena_bytes);
    if (er != hipSuccess) {
        (void)hipFree(h->arena);
        delete h;
        return fail(OCM_EHIP, "hipMemset of the parameter arena failed: %s", hipGetErrorString(er));
    }
    *out = h;
    return OCM_OK;
}

extern "C" void ocm_swin_destroy(ocm_swin_t *h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    delete h;
}

extern "C" int ocm_swin_set_param(ocm_swin_t *h, const char *name, const float *dev_src, size_t count, void *stream) {
    if (!h || !name || !dev_src) return fail(OCM_EINVAL, "ocm_swin_set_param: null argument");
    hipStream_t s = (hipStream_t)stream;
    for (Slot &sl : h->slots) {
        if (sl.name != name) continue;
        if (count != sl.count) return fail(OCM_EINVAL, "%s: expected %zu elements, got %zu", name, sl.count, count);
        char *dst = h->arena + sl.offset;
        if (sl.kind == S_F32) {
            HIP_TRY(hipMemcpyAsync(dst, dev_src, count * 4, hipMemcpyDeviceToDevice, s));
        } else if (sl.kind == S_MAT) {
            HIP_TRY(launch_cast_pad(h->prec, dev_src, dst + sl.row_offset * sl.Kp * h->esz, sl.rows, sl.K, sl.Kp, s));
        } else {
            HIP_TRY(launch_swin_bias_perm(dev_src, (float *)dst, (float *)(h->arena + sl.dense_offset), sl.heads,
                                          h->cfg.window_size, s));
        }
        sl.set = true;
        return OCM_OK;
    }
    return fail(OCM_ENAME, "unknown Swin parameter '%s'", name);
}

extern "C" int ocm_swin_params_ready(const ocm_swin_t *h) {
    if (!h) return fail(OCM_EINVAL, "null handle");
    for (const Slot &sl : h->slots)
        if (!sl.set) return fail(OCM_ESTATE, "parameter '%s' has not been set", sl.name.c_str());
    return OCM_OK;
}

namespace {
// nn.Linear on operand rows of stride lda / output rows of stride ldo. Split-bf16 rows are never padded (every channel
// count of the model is a multiple of the 32-element group), so that mode takes the dense launcher.
hipError_t swin_linear(int prec, const void *a, int64_t lda, const void *w, const float *bias, const float *resid,
                       void *out, int64_t ldo, int M, int N, int K, int epilogue, hipStream_t s) {
    if (prec == 2) {
        if (lda != K || ldo != N) return hipErrorInvalidValue;
        return launch_linear(prec, a, w, bias, resid, out, M, N, K, epilogue, s);
    }
    return launch_linear_ld(prec, a, lda, w, bias, resid, out, ldo, M, N, K, epilogue, s);
}

struct SwinWs {
    float *x, *x2;  // residual stream ping-pong (patch merging writes the other one)
    void *xn, *qkv, *ctx, *hid;
    void *xnp;   // operand rows on the padded grid (stages whose grid is not a multiple of the window)
    float *yp;   // the attention half's fp32 output on the padded grid
    size_t bytes;
};

SwinWs carve_swin(const ocm_swin *h, int batch, char *base) {
    const int ws = h->cfg.window_size;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    // per-stage sizes: real tokens T, tokens on the grid padded to the window Tp, merged tokens T4 (ceil: odd grids are padded)
    size_t x_b = 0, x2_b = 0, xn_b = 0, qkv_b = 0, ctx_b = 0, hid_b = 0, xnp_b = 0, yp_b = 0;
    int H = h->cfg.image_size / 4;
    for (int st = 0; st < h->cfg.num_stages; ++st) {
        const int C = h->chans(st), M = (int)(h->cfg.mlp_ratio * C), Kc = h->Kp(C), Km = h->Kp(M);
        const int Hp = round_up(H, ws);
        const size_t T = (size_t)batch * H * H, Tp = (size_t)batch * Hp * Hp;
        const bool pad = Hp != H;
        auto up = [](size_t &a, size_t b) { a = b > a ? b : a; };
        up(st % 2 ? x2_b : x_b, T * C * 4);
        up(xn_b, T * Kc * h->esz);
        up(qkv_b, (pad ? Tp : T) * 3 * C * h->esz);
        up(ctx_b, (pad ? Tp : T) * Kc * h->esz);
        up(hid_b, T * Km * h->esz);
        if (pad) {
            up(xnp_b, Tp * Kc * h->esz);
            up(yp_b, Tp * C * 4);
        }
        if (st + 1 < h->cfg.num_stages) {
            const int Hn = (H + 1) / 2;
            up(xn_b, (size_t)batch * Hn * Hn * h->Kp(4 * C) * h->esz);  // patch merging: LayerNorm(4C) rows
            H = Hn;
        }
    }
    SwinWs w;
    w.x = (float *)take(x_b);
    w.x2 = (float *)take(x2_b);
    w.xn = take(xn_b);
    w.qkv = take(qkv_b);
    w.ctx = take(ctx_b);
    w.hid = take(hid_b);
    w.xnp = take(xnp_b);
    w.yp = (float *)take(yp_b);
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" size_t ocm_swin_workspace_bytes(const ocm_swin_t *h, int32_t batch) {
    if (!h || batch <= 0) return 0;
    return carve_swin(h, batch, nullptr).bytes;
}

extern "C" int ocm_swin_forward(ocm_swin_t *h, const float *pixel_values, int32_t batch, float *logits, float *pooled,
                                float *last_hidden, void *workspace, size_t workspace_bytes, void *stream) {
    if (!h || !pixel_values || !logits) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0) return fail(OCM_EINVAL, "batch %d must be positive", batch);
    int rc = ocm_swin_params_ready(h);
    if (rc) return rc;
    if (((uintptr_t)pixel_values & 3) || !workspace || ((uintptr_t)workspace & 255))
        return fail(OCM_EINVAL, "workspace must be 256-byte aligned");
    const SwinWs w = carve_swin(h, batch, (char *)workspace);
    if (workspace_bytes < w.bytes) return fail(OCM_EINVAL, "workspace too small: %zu < %zu", workspace_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    const ocm_swin_config &c = h->cfg;
    const int pc = h->prec, ws = c.window_size;
    const float eps = c.ln_eps;
    int H = c.image_size / 4;
    {
        PROF(OCM_K_PATCH, s);
        HIP_TRY(launch_swin_embed(pixel_values, h->ptr<float>(h->emb_w), h->ptr<float>(h->emb_b), h->ptr<float>(h->emb_g),
                                  h->ptr<float>(h->emb_be), w.x, batch, c.num_channels, c.image_size, c.embed_dim, 1e-5f, s));
    }
    float *x = w.x, *xo = w.x2;
    for (int st = 0; st < c.num_stages; ++st) {
        const int C = h->chans(st), heads = c.num_heads[st], M = (int)(c.mlp_ratio * C);
        const int Kc = h->Kp(C), Km = h->Kp(M);
        const size_t T = (size_t)batch * H * H;
        if (T > 0x7fffffff) return fail(OCM_EINVAL, "too many tokens");
        if (Kc != C) HIP_TRY(hipMemsetAsync(w.ctx, 0, T * Kc * h->esz, s));  // K padding of the o_proj operand stays zero
        if (Km != M) HIP_TRY(hipMemsetAsync(w.hid, 0, T * Km * h->esz, s));
        for (size_t b = 0; b < h->stages[st].layers.size(); ++b) {
            const LayerP &lp = h->stages[st].layers[b];
            const int shift = (b % 2 == 1 && H > ws) ? ws / 2 : 0;  // set_shift_and_window_size :576-582
            const int Hp = round_up(H, ws);
            if (Hp != H) {
                // maybe_pad: LayerNorm rows -> the padded grid (zeros at the padded positions), projections and window attention on
                // Hp x Hp tokens, the half's output added back at the real positions (unfused chain, any precision)
                const size_t Tp = (size_t)batch * Hp * Hp;
                if (Tp > 0x7fffffff) return fail(OCM_EINVAL, "too many tokens");
                {
                    PROF(OCM_K_LN, s);
                    HIP_TRY(launch_swin_ln(pc, x, h->ptr<float>(lp.ln1_g), h->ptr<float>(lp.ln1_b), w.xn, T, C, Kc, eps, false, 0, 0, s));
                    HIP_TRY(launch_swin_pad_rows(w.xn, w.xnp, batch, H, H, Hp, Hp, (size_t)Kc * h->esz, s));
                }
                {
                    PROF(OCM_K_QKV, s);
                    HIP_TRY(swin_linear(pc, w.xnp, Kc, h->ptr<char>(lp.wqkv), h->ptr<float>(lp.bqkv), nullptr, w.qkv, 3 * C, (int)Tp,
                                             3 * C, Kc, OCM_EPI_BIAS_BF16, s));
                }
                {
                    PROF(OCM_K_ATTN, s);
                    if (Kc != C) HIP_TRY(hipMemsetAsync(w.ctx, 0, Tp * Kc * h->esz, s));
                    HIP_TRY(launch_swin_window_attention(pc, w.qkv, 3 * C, w.ctx, Kc, h->ptr<float>(lp.bias_perm),
                                                         h->ptr<float>(lp.bias_dense), batch, Hp, Hp, ws, shift, heads, s));
                }
                {
                    PROF(OCM_K_PROJ, s);
                    HIP_TRY(swin_linear(pc, w.ctx, Kc, h->ptr<char>(lp.wo), h->ptr<float>(lp.bo), nullptr, w.yp, C, (int)Tp, C, Kc,
                                             OCM_EPI_BIAS_F32, s));
                    HIP_TRY(launch_swin_crop_add(x, w.yp, batch, H, H, Hp, Hp, C, s));
                }
            } else
            // kernel classes for ocm_prof_begin / ocm_prof_end (tools/bench_swin.py): LayerNorm kernels -> layernorm; the q|k|v
            // projection (fused with layernorm_before or not) -> qkv_gemm; window attention -> attention; attention.output.dense
            // and the patch-merging reduction -> proj_gemm; mlp.fc1 (and the fused LayerNorm + MLP kernel) -> fc1_gemm; mlp.fc2 -> fc2_gemm
            if (h->fuse_mlp && swin_attn_block_fused_supported(pc, C, heads, ws)) {
                // stages 0 and 1: layernorm_before, q | k | v and window attention in ONE kernel — no q | k | v tensor exists; at
                // 96 channels also o_proj and the residual (x read twice, written once), at 192 the context pairs go to the
                // o_proj GEMM (profiled under the attention class)
                {
                    PROF(OCM_K_ATTN, s);
                    HIP_TRY(launch_swin_attn_block(pc, x, h->ptr<float>(lp.ln1_g), h->ptr<float>(lp.ln1_b), h->ptr<char>(lp.wqkv),
                                                   h->ptr<float>(lp.bqkv), h->ptr<char>(lp.wo), h->ptr<float>(lp.bo),
                                                   h->ptr<float>(lp.bias_perm), w.ctx, batch, H, H, ws, shift, heads, C, eps, s));
                }
                if (!swin_attn_block_proj_fused(C)) {
                    PROF(OCM_K_PROJ, s);
                    HIP_TRY(swin_linear(pc, w.ctx, Kc, h->ptr<char>(lp.wo), h->ptr<float>(lp.bo), x, x, C, (int)T, C, Kc,
                                             OCM_EPI_BIAS_RESID_F32, s));
                }
            } else {
            if (h->fuse_mlp && swin_lnqkv_fused_supported(pc, C)) {  // narrow stages: no normalised copy of x in HBM
                PROF(OCM_K_QKV, s);
                HIP_TRY(launch_swin_lnqkv(pc, x, h->ptr<float>(lp.ln1_g), h->ptr<float>(lp.ln1_b), h->ptr<char>(lp.wqkv),
                                          h->ptr<float>(lp.bqkv), w.qkv, T, C, eps, s));
            } else {
                {
                    PROF(OCM_K_LN, s);
                    HIP_TRY(launch_swin_ln(pc, x, h->ptr<float>(lp.ln1_g), h->ptr<float>(lp.ln1_b), w.xn, T, C, Kc, eps,
                                           false, 0, 0, s));
                }
                PROF(OCM_K_QKV, s);
                HIP_TRY(swin_linear(pc, w.xn, Kc, h->ptr<char>(lp.wqkv), h->ptr<float>(lp.bqkv), nullptr, w.qkv, 3 * C,
                                         (int)T, 3 * C, Kc, OCM_EPI_BIAS_BF16, s));
            }
            {
                PROF(OCM_K_ATTN, s);
                HIP_TRY(launch_swin_window_attention(pc, w.qkv, 3 * C, w.ctx, Kc, h->ptr<float>(lp.bias_perm),
                                                     h->ptr<float>(lp.bias_dense), batch, H, H, ws, shift, heads, s));
            }
            {
                PROF(OCM_K_PROJ, s);
                HIP_TRY(swin_linear(pc, w.ctx, Kc, h->ptr<char>(lp.wo), h->ptr<float>(lp.bo), x, x, C, (int)T, C, Kc,
                                         OCM_EPI_BIAS_RESID_F32, s));
            }
            }
            if (h->fuse_mlp && swin_mlp_fused_supported(pc, C, M)) {  // narrow stages: the hidden activations stay on chip
                PROF(OCM_K_FC1, s);
                HIP_TRY(launch_swin_mlp(pc, x, h->ptr<float>(lp.ln2_g), h->ptr<float>(lp.ln2_b), h->ptr<char>(lp.w1),
                                        h->ptr<float>(lp.b1), h->ptr<char>(lp.w2), h->ptr<float>(lp.b2), T, C, M, eps, s));
                continue;
            }
            if (h->fuse_mlp && swin_lnqkv_fused_supported(pc, C)) {  // C = 192: layernorm_after + fc1 + GELU in one kernel
                PROF(OCM_K_FC1, s);
                HIP_TRY(launch_swin_lnlinear(pc, x, h->ptr<float>(lp.ln2_g), h->ptr<float>(lp.ln2_b), h->ptr<char>(lp.w1),
                                             h->ptr<float>(lp.b1), w.hid, T, C, M, true, eps, s));
            } else {
                {
                    PROF(OCM_K_LN, s);
                    HIP_TRY(launch_swin_ln(pc, x, h->ptr<float>(lp.ln2_g), h->ptr<float>(lp.ln2_b), w.xn, T, C, Kc, eps,
                                           false, 0, 0, s));
                }
                PROF(OCM_K_FC1, s);
                HIP_TRY(swin_linear(pc, w.xn, Kc, h->ptr<char>(lp.w1), h->ptr<float>(lp.b1), nullptr, w.hid, Km, (int)T, M,
                                         Kc, OCM_EPI_BIAS_GELU_BF16, s));
            }
            PROF(OCM_K_FC2, s);
            HIP_TRY(swin_linear(pc, w.hid, Km, h->ptr<char>(lp.w2), h->ptr<float>(lp.b2), x, x, C, (int)T, C, Km,
                                     OCM_EPI_BIAS_RESID_F32, s));
        }
        if (st + 1 < c.num_stages) {  // SwinPatchMerging :309-326
            const StageP &sp = h->stages[st];
            const int Hn = (H + 1) / 2;  // SwinPatchMerging.maybe_pad: odd grids gain a row / column of zeros
            const size_t T4 = (size_t)batch * Hn * Hn;
            const int K4 = h->Kp(4 * C);
            {
                PROF(OCM_K_LN, s);
                HIP_TRY(launch_swin_ln(pc, x, h->ptr<float>(sp.red_g), h->ptr<float>(sp.red_b), w.xn, T4, 4 * C, K4, 1e-5f, true,
                                       H, H, s));
            }
            {
                PROF(OCM_K_PROJ, s);
                HIP_TRY(swin_linear(pc, w.xn, K4, h->ptr<char>(sp.red_w), nullptr, nullptr, xo, 2 * C, (int)T4, 2 * C, K4,
                                         OCM_EPI_BIAS_F32, s));
            }
            float *t = x;
            x = xo;
            xo = t;
            H = Hn;
        }
    }
    const int Cl = h->chans(c.num_stages - 1);
    PROF(OCM_K_LN, s);
    HIP_TRY(launch_swin_pool_head(x, h->ptr<float>(h->fin_g), h->ptr<float>(h->fin_b), h->ptr<float>(h->cls_w),
                                  h->ptr<float>(h->cls_b), logits, pooled, last_hidden, batch, H * H, Cl, c.num_labels, eps,
                                  s));
    return OCM_OK;
}

extern "C" int ocm_swin_set_option(ocm_swin_t *h, int32_t option, int32_t value) {
    if (!h) return fail(OCM_EINVAL, "null handle");
    if (option != OCM_SWIN_OPT_FUSE_MLP) return fail(OCM_EINVAL, "unknown Swin option %d", option);
    if (value != 0 && value != 1) return fail(OCM_EINVAL, "OCM_SWIN_OPT_FUSE_MLP takes 0 or 1, got %d", value);
    h->fuse_mlp = value != 0;
    return OCM_OK;
}

extern "C" int ocm_op_swin_lnqkv(int32_t precision, const float *x, const float *gamma, const float *beta, const void *w,
                                 const float *bias, void *qkv, int64_t tokens, int32_t channels, float eps, void *stream) {
    if (!x || !gamma || !beta || !w || !bias || !qkv) return fail(OCM_EINVAL, "null argument");
    const int pc = precision == OCM_PREC_FP32 ? 1 : precision == OCM_PREC_BF16X3 ? 2 : precision == OCM_PREC_BF16 ? 0 : -1;
    if (pc < 0) return fail(OCM_EINVAL, "bad precision");
    if (tokens <= 0 || tokens > 0x7fffffffLL) return fail(OCM_EINVAL, "bad token count %lld", (long long)tokens);
    if (!swin_lnqkv_fused_supported(pc, channels))
        return fail(OCM_EINVAL, "the fused LayerNorm + qkv projection is built for split-bf16 operands and 96, 128 or 192 channels "
                                "(got precision %d, %d channels)", precision, channels);
    HIP_TRY(launch_swin_lnqkv(pc, x, gamma, beta, w, bias, qkv, (size_t)tokens, channels, eps, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_swin_mlp(int32_t precision, float *x, const float *gamma, const float *beta, const void *w1,
                               const float *b1, const void *w2, const float *b2, int64_t tokens, int32_t channels,
                               int32_t hidden, float eps, void *stream) {
    if (!x || !gamma || !beta || !w1 || !b1 || !w2 || !b2) return fail(OCM_EINVAL, "null argument");
    const int pc = precision == OCM_PREC_FP32 ? 1 : precision == OCM_PREC_BF16X3 ? 2 : precision == OCM_PREC_BF16 ? 0 : -1;
    if (pc < 0) return fail(OCM_EINVAL, "bad precision");
    if (tokens <= 0 || tokens > 0x7fffffffLL) return fail(OCM_EINVAL, "bad token count %lld", (long long)tokens);
    if (!swin_mlp_fused_supported(pc, channels, hidden))
        return fail(OCM_EINVAL, "the fused MLP is built for split-bf16 operands, channels 96 or 128 and hidden = 4 x channels "
                                "(got precision %d, %d, %d)", precision, channels, hidden);
    HIP_TRY(launch_swin_mlp(pc, x, gamma, beta, w1, b1, w2, b2, (size_t)tokens, channels, hidden, eps, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_swin_attn_block(int32_t precision, float *x, const float *gamma, const float *beta, const void *wqkv,
                                      const float *bqkv, const void *wo, const float *bo, const float *rel_table, float *scratch,
                                      int32_t batch, int32_t height, int32_t width, int32_t window, int32_t shift, int32_t heads,
                                      float eps, void *stream) {
    if (!x || !gamma || !beta || !wqkv || !bqkv || !wo || !bo || !rel_table || !scratch) return fail(OCM_EINVAL, "null argument");
    const int pc = precision == OCM_PREC_FP32 ? 1 : precision == OCM_PREC_BF16X3 ? 2 : precision == OCM_PREC_BF16 ? 0 : -1;
    if (pc < 0) return fail(OCM_EINVAL, "bad precision");
    if (window < 2 || window > 7 || height <= 0 || width <= 0 || height % window || width % window || batch <= 0 || heads <= 0 ||
        shift < 0 || shift >= window)
        return fail(OCM_EINVAL, "bad window geometry");
    const int Cn = heads * 32;
    if (!swin_attn_block_fused_supported(pc, Cn, heads, window))
        return fail(OCM_EINVAL, "the fused attention half is built for split-bf16 operands and 3, 4 or 6 heads of 32 channels "
                                "(got precision %d, %d heads)", precision, heads);
    hipStream_t s = (hipStream_t)stream;
    const int64_t T = (int64_t)batch * height * width;
    if (T > 0x7fffffffLL) return fail(OCM_EINVAL, "too many tokens");
    void *ctx = scratch + (size_t)heads * 4096;  // context pairs (6 heads: o_proj is a GEMM of its own)
    HIP_TRY(launch_swin_bias_perm(rel_table, scratch, nullptr, heads, window, s));
    HIP_TRY(launch_swin_attn_block(pc, x, gamma, beta, wqkv, bqkv, wo, bo, scratch, ctx, batch, height, width, window, shift,
                                   heads, Cn, eps, s));
    if (!swin_attn_block_proj_fused(Cn))
        HIP_TRY(swin_linear(pc, ctx, Cn, wo, bo, x, x, Cn, (int)T, Cn, Cn, OCM_EPI_BIAS_RESID_F32, s));
    return OCM_OK;
}

extern "C" int ocm_op_swin_window_attention(int32_t precision, const void *qkv, int32_t ld, void *ctx, int32_t ldc,
                                            const float *rel_table, float *scratch, int32_t batch, int32_t height,
                                            int32_t width, int32_t window, int32_t shift, int32_t heads, void *stream) {
    if (!qkv || !ctx || !rel_table || !scratch) return fail(OCM_EINVAL, "null argument");
    if (precision != OCM_PREC_BF16 && precision != OCM_PREC_FP32 && precision != OCM_PREC_BF16X3)
        return fail(OCM_EINVAL, "bad precision");
    if (precision == OCM_PREC_BF16X3 && (ld % 32 || ldc % 32))
        return fail(OCM_EINVAL, "split-bf16 rows are whole groups of 32 elements: ld %d, ldc %d", ld, ldc);
    if (window < 2 || window > 7 || height % window || width % window || batch <= 0 || heads <= 0 || shift < 0 ||
        shift >= window)
        return fail(OCM_EINVAL, "bad window geometry");
    hipStream_t s = (hipStream_t)stream;
    float *perm = scratch, *dense = scratch + (size_t)heads * 4096;
    HIP_TRY(launch_swin_bias_perm(rel_table, perm, dense, heads, window, s));
    HIP_TRY(launch_swin_window_attention(precision == OCM_PREC_FP32 ? 1 : precision == OCM_PREC_BF16X3 ? 2 : 0, qkv, ld, ctx, ldc, perm, dense, batch, height,
                                         width, window, shift, heads, s));
    return OCM_OK;
}
