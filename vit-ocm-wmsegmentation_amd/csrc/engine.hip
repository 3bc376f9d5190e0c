// engine.hip — the C ABI of include/ocm_vit.h: parameter store, workspace carving and the
// launch sequence of one ViT forward (dino/vision_transformer.py:198-256 of the reference).
// Host code only; every kernel lives in kernels_*.hip. Nothing here synchronises the device.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "host_common.h"
#include "launch.h"
#include "dev_knobs.h"

static thread_local std::string g_err;

int ocm_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define fail ocm_fail

// ------------------------------------------------------------------------------------------
// kernel-class timing with hipEvents (diagnostic)
// ------------------------------------------------------------------------------------------
Prof g_prof;  // host_common.h: the session state shared by engine.hip and swin_engine.hip

extern "C" int ocm_prof_begin(uint32_t class_mask, int32_t max_launches) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.on.load(std::memory_order_relaxed)) return fail(OCM_ESTATE, "profiling already active");
    if (max_launches <= 0) return fail(OCM_EINVAL, "max_launches must be positive");
    g_prof.ev.resize((size_t)max_launches * 2);
    g_prof.cls.assign(max_launches, 0);
    for (auto &e : g_prof.ev) HIP_TRY(hipEventCreate(&e));
    g_prof.used = 0;
    g_prof.mask = class_mask;
    ++g_prof.gen;
    g_prof.on.store(true, std::memory_order_release);
    return OCM_OK;
}

extern "C" int ocm_prof_end(double *ms_per_class, int64_t *launches_per_class) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on.load(std::memory_order_relaxed)) return fail(OCM_ESTATE, "profiling is not active");
    g_prof.on.store(false, std::memory_order_release);
    for (int c = 0; c < OCM_K_COUNT; ++c) {
        if (ms_per_class) ms_per_class[c] = 0.0;
        if (launches_per_class) launches_per_class[c] = 0;
    }
    int rc = OCM_OK;
    for (size_t i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) {
            rc = fail(OCM_EHIP, "event timing failed: %s", hipGetErrorString(e));
            break;
        }
        if (ms_per_class) ms_per_class[g_prof.cls[i]] += ms;
        if (launches_per_class) launches_per_class[g_prof.cls[i]] += 1;
    }
    for (auto &e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear();
    g_prof.cls.clear();
    g_prof.used = 0;
    return rc;
}

#ifdef OCM_DEV  // development build only: include/ocm_vit_dev.h
extern "C" int ocm_debug_knob(int32_t which, int32_t value) {
    if (which < 0 || which >= 8) return fail(OCM_EINVAL, "knob %d out of range", which);
    g_ocm_knobs[which] = value;
    return OCM_OK;
}
#endif

extern "C" int ocm_abi_version(void) { return OCM_ABI_VERSION; }
extern "C" const char *ocm_last_error(void) { return g_err.c_str(); }
extern "C" int32_t ocm_n_pad(int32_t n_tokens) { return ocm_round_up(n_tokens, 8); }
extern "C" int32_t ocm_n_pad_prec(int32_t precision, int32_t n_tokens) { return ocm_n_pad_for(precision, n_tokens); }

// ------------------------------------------------------------------------------------------
// parameter store
// ------------------------------------------------------------------------------------------
enum ParamKind { P_F32, P_BF16, P_PATCH };
struct Param {
    std::string name;
    ParamKind kind;
    size_t count;   // elements in the reference tensor
    size_t offset;  // bytes into the arena
    bool set;
    bool optional = false;  // not required by ocm_vit_params_ready (mask_token)
};

struct BlockP {
    int ln1_g, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_g, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b;
    // split-bf16 engines only (LayerNorm folded into its consumer): fp32 copies of the two weight matrices
    // that follow a LayerNorm, and what is derived from them: W' = W * gamma as split pairs, c = row sums of W',
    // d = W beta + bias
    int qkv_w32 = -1, fc1_w32 = -1, qkv_wf = -1, fc1_wf = -1, qkv_c = -1, qkv_d = -1, fc1_c = -1, fc1_d = -1;
    bool fold_dirty = true;
};

struct ocm_vit {
    ocm_vit_config cfg;
    int D, H, L, M, p, C, Kpe;
    int hd;    // head_dim: 64 (and 128 in split-bf16 precision, model.py:96-97) runs the MFMA attention kernels, anything else the
               // generic fp32 attention
    bool mfma_heads() const { return hd == 64 || hd == 128; }  // (128-wide heads: every precision since round 4)
    int prec;  // 0 = bf16 operands, 1 = fp32 operands, 2 = split-bf16 pairs (element size esz of matrices / activations)
    size_t esz;
    std::vector<Param> params;
    std::vector<BlockP> blk;
    int cls, pe_w, pe_b, norm_g, norm_b, mask_tok;
    char *arena;
    size_t arena_bytes;
    // OCM_USE_GRAPH: the launch sequence of the last forward as an instantiated hipGraph, keyed on every value
    // the kernels' arguments were derived from (shapes, flags, pointers)
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_stream = nullptr;  // stands in for the (uncapturable) legacy default stream
    ocm_vit_io graph_key;
    bool graph_valid = false;
    uint64_t graph_hits = 0, graph_captures = 0;
    int32_t opt[OCM_OPT_COUNT] = {0};  // ocm_vit_set_option: per-handle dispatch options (0 = automatic)
    bool can_fold() const { return prec == 2; }  // LayerNorm folding exists for the split-bf16 kernels
    bool folding() const { return can_fold() && opt[OCM_OPT_FOLD_LN] != 1; }

    int add(const std::string &name, ParamKind kind, size_t count, size_t stored_elems) {
        Param pr{name, kind, count, arena_bytes, false};
        const size_t bytes = stored_elems * (kind == P_F32 ? 4 : esz);
        arena_bytes += (bytes + 255) & ~(size_t)255;
        params.push_back(pr);
        return (int)params.size() - 1;
    }
    template <class T>
    T *ptr(int idx) const {
        return (T *)(arena + params[idx].offset);
    }
};

extern "C" int ocm_vit_create(const ocm_vit_config *cfg, ocm_vit_t **out) {
    if (!cfg || !out) return fail(OCM_EINVAL, "ocm_vit_create: null argument");
    const int D = cfg->embed_dim, H = cfg->num_heads, p = cfg->patch_size, C = cfg->in_chans;
    if (D <= 0 || D % 64 || D > 1024) return fail(OCM_EINVAL, "embed_dim %d must be a multiple of 64, <= 1024", D);
    if (H <= 0 || D % H || (D / H) % 8 || D / H > 512)
        return fail(OCM_EINVAL, "head_dim = embed_dim / num_heads must be a multiple of 8, <= 512 (embed_dim %d, num_heads %d)", D, H);
    if (p < 8 || p % 8 || p > 32) return fail(OCM_EINVAL, "patch_size %d must be 8, 16, 24 or 32", p);
    if (C != 1 && C != 3) return fail(OCM_EINVAL, "in_chans %d must be 1 or 3", C);
    if ((C * p * p) % 64) return fail(OCM_EINVAL, "in_chans*patch_size^2 = %d must be a multiple of 64", C * p * p);
    if (cfg->mlp_hidden <= 0 || cfg->mlp_hidden % 64)
        return fail(OCM_EINVAL, "mlp_hidden %d must be a multiple of 64", cfg->mlp_hidden);
    if (cfg->depth <= 0) return fail(OCM_EINVAL, "depth %d must be positive", cfg->depth);
    if (cfg->precision != OCM_PREC_BF16 && cfg->precision != OCM_PREC_FP32 && cfg->precision != OCM_PREC_BF16X3)
        return fail(OCM_EINVAL, "precision %d is not one of OCM_PREC_BF16 / OCM_PREC_FP32 / OCM_PREC_BF16X3", cfg->precision);
    ocm_vit *h = new ocm_vit();
    h->cfg = *cfg;
    h->hd = D / H;
    h->D = D; h->H = H; h->L = cfg->depth; h->M = cfg->mlp_hidden; h->p = p; h->C = C; h->Kpe = C * p * p;
    h->prec = cfg->precision;  // OCM_PREC_* values are the kernels' `prec` selector
    h->esz = h->prec ? 4 : 2;
    h->arena = nullptr;
    h->arena_bytes = 0;
    const size_t d = D, m = cfg->mlp_hidden;
    h->cls = h->add("cls_token", P_F32, d, d);
    // the reference conv weight is (D, C_ref, p, p); it is stored as bf16 [D][C*p*p] (C_ref folded if needed)
    h->pe_w = h->add("patch_embed.proj.weight", P_PATCH, 0, d * h->Kpe);
    h->pe_b = h->add("patch_embed.proj.bias", P_F32, d, d);
    for (int i = 0; i < h->L; ++i) {
        const std::string b = "blocks." + std::to_string(i) + ".";
        BlockP bp;
        bp.ln1_g = h->add(b + "norm1.weight", P_F32, d, d);
        bp.ln1_b = h->add(b + "norm1.bias", P_F32, d, d);
        bp.qkv_w = h->add(b + "attn.qkv.weight", P_BF16, 3 * d * d, 3 * d * d);
        bp.qkv_b = h->add(b + "attn.qkv.bias", P_F32, 3 * d, 3 * d);
        bp.proj_w = h->add(b + "attn.proj.weight", P_BF16, d * d, d * d);
        bp.proj_b = h->add(b + "attn.proj.bias", P_F32, d, d);
        bp.ln2_g = h->add(b + "norm2.weight", P_F32, d, d);
        bp.ln2_b = h->add(b + "norm2.bias", P_F32, d, d);
        bp.fc1_w = h->add(b + "mlp.fc1.weight", P_BF16, m * d, m * d);
        bp.fc1_b = h->add(b + "mlp.fc1.bias", P_F32, m, m);
        bp.fc2_w = h->add(b + "mlp.fc2.weight", P_BF16, d * m, d * m);
        bp.fc2_b = h->add(b + "mlp.fc2.bias", P_F32, d, d);
        if (h->prec == 2) {  // derived / shadow entries: never set through ocm_vit_set_param by name
            auto derived = [&](const char *nm, ParamKind kind, size_t elems) {
                const int idx = h->add(b + nm, kind, elems, elems);
                h->params[idx].optional = true;
                return idx;
            };
            bp.qkv_w32 = derived("~qkv.w32", P_F32, 3 * d * d);
            bp.fc1_w32 = derived("~fc1.w32", P_F32, m * d);
            bp.qkv_wf = derived("~qkv.wf", P_BF16, 3 * d * d);
            bp.fc1_wf = derived("~fc1.wf", P_BF16, m * d);
            bp.qkv_c = derived("~qkv.c", P_F32, 3 * d);
            bp.qkv_d = derived("~qkv.d", P_F32, 3 * d);
            bp.fc1_c = derived("~fc1.c", P_F32, m);
            bp.fc1_d = derived("~fc1.d", P_F32, m);
        }
        h->blk.push_back(bp);
    }
    h->norm_g = h->add("norm.weight", P_F32, d, d);
    h->norm_b = h->add("norm.bias", P_F32, d, d);
    h->mask_tok = h->add("mask_token", P_F32, d, d);
    h->params[h->mask_tok].optional = true;
    hipError_t e = hipMalloc((void **)&h->arena, h->arena_bytes);
    if (e != hipSuccess) {
        const size_t bytes = h->arena_bytes;
        delete h;
        return fail(OCM_ENOMEM, "hipMalloc(%zu) for parameters failed: %s", bytes, hipGetErrorString(e));
    }
    *out = h;
    return OCM_OK;
}

extern "C" void ocm_vit_destroy(ocm_vit_t *h) {
    if (!h) return;
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph_stream) (void)hipStreamDestroy(h->graph_stream);
    if (h->arena) (void)hipFree(h->arena);
    delete h;
}

extern "C" int ocm_vit_set_param(ocm_vit_t *h, const char *name, const float *dev_src, size_t count, void *stream) {
    if (!h || !name || !dev_src) return fail(OCM_EINVAL, "ocm_vit_set_param: null argument");
    hipStream_t s = (hipStream_t)stream;
    for (Param &pr : h->params) {
        if (pr.name != name) continue;
        char *dst = h->arena + pr.offset;
        if (pr.kind == P_F32) {
            if (count != pr.count) return fail(OCM_EINVAL, "%s: expected %zu elements, got %zu", name, pr.count, count);
            HIP_TRY(hipMemcpyAsync(dst, dev_src, count * 4, hipMemcpyDeviceToDevice, s));
        } else if (pr.kind == P_BF16) {  // matrix: bf16 copy, or fp32 copy in OCM_PREC_FP32
            if (count != pr.count) return fail(OCM_EINVAL, "%s: expected %zu elements, got %zu", name, pr.count, count);
            if (h->prec == 2)
                HIP_TRY(launch_cast_split(dev_src, dst, count, s));
            else if (h->prec)
                HIP_TRY(hipMemcpyAsync(dst, dev_src, count * 4, hipMemcpyDeviceToDevice, s));
            else
                HIP_TRY(launch_cast_bf16(dev_src, (bf16 *)dst, count, s));
        } else {  // conv weight (D, C_ref, p, p)
            const size_t pp = (size_t)h->p * h->p, per = (size_t)h->D * pp;
            if (count == 0 || count % per) return fail(OCM_EINVAL, "%s: %zu elements is not (D=%d, C, %d, %d)", name, count, h->D, h->p, h->p);
            const int cref = (int)(count / per);
            if (cref == h->C) {
                if (h->prec == 2)
                    HIP_TRY(launch_cast_split(dev_src, dst, count, s));
                else if (h->prec)
                    HIP_TRY(hipMemcpyAsync(dst, dev_src, count * 4, hipMemcpyDeviceToDevice, s));
                else
                    HIP_TRY(launch_cast_bf16(dev_src, (bf16 *)dst, count, s));
            } else if (h->C == 1) {  // grayscale fold: W_eff = W.sum(dim=1)
                if (h->prec == 2)
                    HIP_TRY(launch_fold_split(dev_src, dst, h->D, cref, (int)pp, s));
                else if (h->prec)
                    HIP_TRY(launch_fold_f32(dev_src, (float *)dst, h->D, cref, (int)pp, s));
                else
                    HIP_TRY(launch_fold_cast_bf16(dev_src, (bf16 *)dst, h->D, cref, (int)pp, s));
            } else {
                return fail(OCM_EINVAL, "%s: weight has %d input channels, engine was created with in_chans=%d", name, cref, h->C);
            }
        }
        pr.set = true;
        if (h->can_fold() && pr.name.compare(0, 7, "blocks.") == 0) {  // keep the folded-LayerNorm operands in step
            const size_t dot = pr.name.find('.', 7);
            const int bi = atoi(pr.name.c_str() + 7);
            if (dot != std::string::npos && bi >= 0 && bi < h->L) {
                BlockP &bp = h->blk[bi];
                const std::string leaf = pr.name.substr(dot + 1);
                if (leaf == "attn.qkv.weight")
                    HIP_TRY(hipMemcpyAsync(h->ptr<char>(bp.qkv_w32), dev_src, count * 4, hipMemcpyDeviceToDevice, s));
                if (leaf == "mlp.fc1.weight")
                    HIP_TRY(hipMemcpyAsync(h->ptr<char>(bp.fc1_w32), dev_src, count * 4, hipMemcpyDeviceToDevice, s));
                if (leaf.compare(0, 4, "norm") == 0 || leaf.compare(0, 8, "attn.qkv") == 0 || leaf.compare(0, 7, "mlp.fc1") == 0)
                    bp.fold_dirty = true;
            }
        }
        return OCM_OK;
    }
    return fail(OCM_ENAME, "unknown parameter '%s'", name);
}

extern "C" int ocm_vit_set_option(ocm_vit_t *h, int32_t option, int32_t value) {
    if (!h) return fail(OCM_EINVAL, "null handle");
    if (option < 0 || option >= OCM_OPT_COUNT) return fail(OCM_EINVAL, "unknown option %d", option);
    if (option == OCM_OPT_FUSE_LN && (value < 0 || value > 2)) return fail(OCM_EINVAL, "OCM_OPT_FUSE_LN takes 0 (auto), 1 (never) or 2 (always)");
    if (option == OCM_OPT_FOLD_LN && (value < 0 || value > 2)) return fail(OCM_EINVAL, "OCM_OPT_FOLD_LN takes 0 (auto), 1 (never) or 2 (always)");
    if (h->opt[option] != value) h->graph_valid = false;  // a cached launch sequence was recorded under the old setting
    h->opt[option] = value;
    return OCM_OK;
}

// (Re)build W' = W * gamma, c and d of every block whose LayerNorm / qkv / fc1 parameters changed since the last forward.
static int ensure_folded(ocm_vit *h, hipStream_t s) {
    if (!h->can_fold()) return OCM_OK;
    for (int i = 0; i < h->L; ++i) {
        BlockP &bp = h->blk[i];
        if (!bp.fold_dirty) continue;
        HIP_TRY(launch_fold_ln(h->ptr<float>(bp.qkv_w32), h->ptr<float>(bp.ln1_g), h->ptr<float>(bp.ln1_b), h->ptr<float>(bp.qkv_b),
                               h->ptr<char>(bp.qkv_wf), h->ptr<float>(bp.qkv_c), h->ptr<float>(bp.qkv_d), 3 * h->D, h->D, s));
        HIP_TRY(launch_fold_ln(h->ptr<float>(bp.fc1_w32), h->ptr<float>(bp.ln2_g), h->ptr<float>(bp.ln2_b), h->ptr<float>(bp.fc1_b),
                               h->ptr<char>(bp.fc1_wf), h->ptr<float>(bp.fc1_c), h->ptr<float>(bp.fc1_d), h->M, h->D, s));
        bp.fold_dirty = false;
    }
    return OCM_OK;
}

extern "C" int ocm_vit_params_ready(const ocm_vit_t *h) {
    if (!h) return fail(OCM_EINVAL, "null handle");
    for (const Param &pr : h->params)
        if (!pr.set && !pr.optional) return fail(OCM_ESTATE, "parameter '%s' has not been set", pr.name.c_str());
    return OCM_OK;
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
struct Workspace {  // E = bf16 (OCM_PREC_BF16) or float (OCM_PREC_FP32)
    float *x;     // [T][D]   fp32 residual stream
    void *xn;     // [T][D] E LayerNorm output (GEMM A operand)
    void *q, *k;  // [B*H][n_pad][hd] E
    void *vt;     // [B*H][hd][n_pad] E
    void *ctx;    // [T][D] E attention output, heads merged
    void *hid;    // [T][M] E GELU(fc1)
    float *lse;   // [B*H][N]
    float *qkv32; // [3][B][H][N][hd] fp32: the qkv tensor of heads that are not 64 wide (then q / k / vt are unused)
    float *kpart; // key-slice partial results of the attention at long sequences and small batch (launch.h), or null
    size_t kpart_bytes;
    float *part;  // [OCM_SPLITK][T][D] fp32 partial sums of mlp.fc2 when it runs as split-K (T <= OCM_SPLITK_MAX_ROWS), else null
    float *stats; // [2][T][D/64][2] row sums (sum x, sum x^2 per 64-column slot) of the residual stream at the LayerNorm site being
                  // produced / consumed (folded LayerNorm: even sites in half 0, odd sites in half 1)
    float *shift; // [2][T] the per-row constants the pairs and sums of that site are centred by (launch.h: row centring), same halves
    float *tok_shift;  // [n] the first site's per-token constants (mean of the positional row + mean of the patch-embedding bias)
    size_t bytes;
};

static Workspace carve(const ocm_vit *h, int batch, int n, char *base) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *pch = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return pch;
    };
    const size_t T = (size_t)batch * n, BH = (size_t)batch * h->H, np = ocm_n_pad_for(h->prec, n);
    const size_t e = h->esz;
    w.x = (float *)take(T * h->D * 4);
    w.xn = take(T * h->D * e);
    w.lse = (float *)take(BH * n * 4);
    w.stats = h->can_fold() ? (float *)take(2 * T * (size_t)(h->D / 64) * 8) : nullptr;
    w.shift = h->can_fold() ? (float *)take(2 * T * 4) : nullptr;
    w.tok_shift = h->can_fold() ? (float *)take((size_t)n * 4) : nullptr;
    w.part = (h->prec == 2 && T <= (size_t)OCM_SPLITK_MAX_ROWS) ? (float *)take((size_t)OCM_SPLITK * T * h->D * 4) : nullptr;
    w.kpart_bytes = attention_ksplit_bytes(h->prec, batch, n, h->H, h->hd);
    w.kpart = w.kpart_bytes ? (float *)take(w.kpart_bytes) : nullptr;
    // The attention operands (q, k, V^T or the fp32 qkv tensor, then the context) are dead once attn.proj has run, and the
    // hidden activations only live from mlp.fc1 to mlp.fc2: the two groups share one region. ViT-S/16 at B = 64 in
    // split-bf16: 123 MB instead of 195 MB, which with the 85 MB of weights keeps a forward's working set inside the
    // 256 MiB Infinity Cache (weights and activations are then re-read from it rather than from HBM).
    const size_t base_off = off;
    w.q = w.k = w.vt = nullptr;
    w.qkv32 = nullptr;
    if (h->mfma_heads()) {
        w.q = take(BH * np * h->hd * e);
        w.k = take(BH * np * h->hd * e);
        w.vt = take(BH * np * h->hd * e);
    } else {
        w.qkv32 = (float *)take(3 * T * h->D * 4);
    }
    w.ctx = take(T * h->D * e);
    const size_t attn_end = off;
    off = base_off;
    w.hid = take(T * h->M * e);
    if (off < attn_end) off = attn_end;
    w.bytes = off;
    return w;
}

extern "C" size_t ocm_vit_workspace_bytes(const ocm_vit_t *h, int32_t batch, int32_t n_tokens) {
    if (!h || batch <= 0 || n_tokens <= 1) return 0;
    return carve(h, batch, n_tokens, nullptr).bytes;
}

// ------------------------------------------------------------------------------------------
// forward pieces
// ------------------------------------------------------------------------------------------
static int check_ws(const ocm_vit *h, int batch, int n, void *ws, size_t ws_bytes) {
    if (!ws) return fail(OCM_ENOMEM, "workspace is null");
    if ((uintptr_t)ws & 255) return fail(OCM_EINVAL, "workspace must be 256-byte aligned");
    const size_t need = carve(h, batch, n, nullptr).bytes;
    if (ws_bytes < need) return fail(OCM_ENOMEM, "workspace too small: %zu < %zu bytes", ws_bytes, need);
    return OCM_OK;
}

// Block.forward (:106-114). x is updated in place unless attn_only.
// `xn_ready`: w.xn already holds norm1(x) (written by the previous block's fused fc2 epilogue). `next_g` / `next_b`: the
// LayerNorm that consumes this block's output next (the next block's norm1) — when given and the embedding width has a
// fused kernel, fc2's epilogue writes it into w.xn and *xn_out is set.
static int run_block(const ocm_vit *h, int i, const Workspace &w, float *x, int batch, int n, bool attn_only,
                     float *out_attn, float *out_qkv, const int32_t *query_rows, int n_rows, float *out_rows,
                     hipStream_t s, bool xn_ready = false, const float *next_g = nullptr, const float *next_b = nullptr,
                     bool *xn_out = nullptr) {
    const BlockP &bp = h->blk[i];
    const int D = h->D, H = h->H, T = batch * n, np = ocm_n_pad_for(h->prec, n);
    const float eps = h->cfg.ln_eps, scale = h->cfg.qk_scale;
    const int pc = h->prec, lnk = ln_kind_of_prec(pc);
    // The fused GEMM + LayerNorm kernel owns full rows (64 x D tiles, D / 128 times the W bytes per workgroup and step):
    // it pays once there are enough row tiles to occupy the chip (measured: +2 % at T = 12 608; at T = 197, four
    // workgroups stream all of W each: 1.52 ms per forward against 1.14 ms). Handle option OCM_OPT_FUSE_LN: 1 = never,
    // 2 = always.
    // Above 512 tiles of 128 x 128 in the (T x D) output the un-fused pair wins again: those tiles fill two workgroups
    // per CU and move half the LDS bytes per MFMA of a 64-row tile (ViT-S/8 slab sweep, T = 48 405: 575 -> 556 ms).
    const long t128 = (long)((T + 127) / 128) * (D / 128);
    const int fuse_opt = h->opt[OCM_OPT_FUSE_LN];
    const bool fuse_ln = linear_resid_ln_supported(D) && fuse_opt != 1 && ((T >= 8192 && t128 < 512) || fuse_opt == 2);
    if (xn_out) *xn_out = false;
    // y = attn(norm1(x))
    if (!xn_ready) { PROF(OCM_K_LN, s); HIP_TRY(launch_layernorm(x, h->ptr<float>(bp.ln1_g), h->ptr<float>(bp.ln1_b), w.xn, lnk, T, D, eps, s)); }
    // a block that stops after its probabilities (get_last_selfattention) and returns no qkv never reads V
    const bool want_v = !(attn_only && !out_qkv);
    if (!h->mfma_heads()) {  // generic heads: fp32 qkv tensor -> fp32 FMA attention
        float *qkv = out_qkv ? out_qkv : w.qkv32;
        { PROF(OCM_K_QKV, s); HIP_TRY(launch_qkv(pc, w.xn, h->ptr<char>(bp.qkv_w), h->ptr<float>(bp.qkv_b), nullptr, nullptr, nullptr, qkv, batch, n, np, H, h->hd, want_v, s)); }
        { PROF(OCM_K_ATTN, s); HIP_TRY(launch_attention_generic(pc, qkv, attn_only ? nullptr : w.ctx, out_attn, query_rows, n_rows, out_rows, batch, n, H, h->hd, scale, s)); }
        if (attn_only) return OCM_OK;
    } else {
    void *vt_dst = want_v ? w.vt : nullptr;
    { PROF(OCM_K_QKV, s); HIP_TRY(launch_qkv(pc, w.xn, h->ptr<char>(bp.qkv_w), h->ptr<float>(bp.qkv_b), w.q, w.k, vt_dst, out_qkv, batch, n, np, H, h->hd, want_v, s)); }
    // selected query rows: a slice of the probabilities when those are materialised for this block anyway,
    // else their own fp32 dot-product kernel (never the (H,N,N) matrix)
    if (out_rows && !out_attn) { PROF(OCM_K_PROBS, s); HIP_TRY(launch_attention_rows(pc, w.q, w.k, query_rows, n_rows, out_rows, batch, n, np, H, scale, s, h->hd)); }
    if (attn_only) {
        if (out_attn) {
            { PROF(OCM_K_ATTN, s); HIP_TRY(launch_attention(pc, w.q, w.k, w.vt, nullptr, w.lse, batch, n, np, H, scale, s, h->hd, w.kpart, w.kpart_bytes)); }
            { PROF(OCM_K_PROBS, s); HIP_TRY(launch_attention_probs(pc, w.q, w.k, w.lse, out_attn, batch, n, np, H, scale, s, h->hd)); }
            if (out_rows) { PROF(OCM_K_PROBS, s); HIP_TRY(launch_rows_from_probs(out_attn, query_rows, n_rows, out_rows, batch, n, H, s)); }
        }
        return OCM_OK;
    }
    { PROF(OCM_K_ATTN, s); HIP_TRY(launch_attention(pc, w.q, w.k, w.vt, w.ctx, out_attn ? w.lse : nullptr, batch, n, np, H, scale, s, h->hd, w.kpart, w.kpart_bytes)); }
    if (out_attn) {
        { PROF(OCM_K_PROBS, s); HIP_TRY(launch_attention_probs(pc, w.q, w.k, w.lse, out_attn, batch, n, np, H, scale, s, h->hd)); }
        if (out_rows) { PROF(OCM_K_PROBS, s); HIP_TRY(launch_rows_from_probs(out_attn, query_rows, n_rows, out_rows, batch, n, H, s)); }
    }
    }
    // x = x + proj(ctx); xn = norm2(x) — one kernel when the workgroup can own full rows
    if (fuse_ln) {
        PROF(OCM_K_PROJ, s);
        HIP_TRY(launch_linear_resid_ln(pc, w.ctx, h->ptr<char>(bp.proj_w), h->ptr<float>(bp.proj_b), x, x, h->ptr<float>(bp.ln2_g), h->ptr<float>(bp.ln2_b), w.xn, T, D, D, eps, s));
    } else {
        { PROF(OCM_K_PROJ, s); HIP_TRY(launch_linear(pc, w.ctx, h->ptr<char>(bp.proj_w), h->ptr<float>(bp.proj_b), x, x, T, D, D, OCM_EPI_BIAS_RESID_F32, s)); }
        { PROF(OCM_K_LN, s); HIP_TRY(launch_layernorm(x, h->ptr<float>(bp.ln2_g), h->ptr<float>(bp.ln2_b), w.xn, lnk, T, D, eps, s)); }
    }
    // x = x + fc2(gelu(fc1(xn))) [; xn = the next block's norm1(x)]
    { PROF(OCM_K_FC1, s); HIP_TRY(launch_linear(pc, w.xn, h->ptr<char>(bp.fc1_w), h->ptr<float>(bp.fc1_b), nullptr, w.hid, T, h->M, D, OCM_EPI_BIAS_GELU_BF16, s)); }
    if (fuse_ln && next_g && next_b) {
        PROF(OCM_K_FC2, s);
        HIP_TRY(launch_linear_resid_ln(pc, w.hid, h->ptr<char>(bp.fc2_w), h->ptr<float>(bp.fc2_b), x, x, next_g, next_b, w.xn, T, D, h->M, eps, s));
        if (xn_out) *xn_out = true;
    } else {
        PROF(OCM_K_FC2, s);
        StatsOut so;
        if (fuse_opt == 0) so.part = w.part;  // few rows: split-K (launch.h); an explicit OCM_OPT_FUSE_LN setting keeps the plain
                                              // kernel, whose accumulation order the fused GEMM + LayerNorm kernel shares
        HIP_TRY(launch_linear(pc, w.hid, h->ptr<char>(bp.fc2_w), h->ptr<float>(bp.fc2_b), x, x, T, D, h->M, OCM_EPI_BIAS_RESID_F32, s, LnFold(), so));
    }
    return OCM_OK;
}

// Block.forward with both LayerNorms folded into the GEMMs that consume them (split-bf16 engines). On entry w.xn holds the
// split pairs of x and stats site 2i its row sums; on exit (unless attn_only) the same for the block's output at site 2i+2.
//   qkv  = rstd1 * (x W1'^T - mu1 c1) + d1          (attn.qkv with norm1 folded)
//   x   += proj(attention)      -> x, split(x), sums at site 2i+1
//   hid  = gelu(rstd2 * (x W2'^T - mu2 c2) + d2)    (mlp.fc1 with norm2 folded)
//   x   += fc2(hid)             -> x, split(x), sums at site 2i+2
// No LayerNorm kernel runs, and the N = D GEMMs are free to use tiles that do not own whole rows.
static int run_block_folded(const ocm_vit *h, int i, const Workspace &w, float *x, int batch, int n, bool attn_only,
                            float *out_attn, float *out_qkv, const int32_t *query_rows, int n_rows, float *out_rows,
                            hipStream_t s) {
    const BlockP &bp = h->blk[i];
    const int D = h->D, H = h->H, T = batch * n, np = ocm_n_pad_for(h->prec, n), pc = h->prec;
    const float scale = h->cfg.qk_scale;
    // sites 2i (block input) and 2i+2 (block output) use half 0 of the statistics buffer, site 2i+1 (after attn.proj) half 1
    float *st_in = w.stats, *st_mid = w.stats + (size_t)T * (D / 64) * 2, *st_out = w.stats;
    LnFold ln1, ln2;
    ln1.stats = st_in, ln1.c = h->ptr<float>(bp.qkv_c), ln1.d = h->ptr<float>(bp.qkv_d), ln1.inv_dim = 1.0f / D, ln1.eps = h->cfg.ln_eps;
    ln2.stats = st_mid, ln2.c = h->ptr<float>(bp.fc1_c), ln2.d = h->ptr<float>(bp.fc1_d), ln2.inv_dim = 1.0f / D, ln2.eps = h->cfg.ln_eps;
    ln1.nslot = ln2.nslot = D / 64;
    const bool want_v = !(attn_only && !out_qkv);
    if (!h->mfma_heads()) {
        float *qkv = out_qkv ? out_qkv : w.qkv32;
        { PROF(OCM_K_QKV, s); HIP_TRY(launch_qkv(pc, w.xn, h->ptr<char>(bp.qkv_wf), nullptr, nullptr, nullptr, nullptr, qkv, batch, n, np, H, h->hd, want_v, s, ln1)); }
        { PROF(OCM_K_ATTN, s); HIP_TRY(launch_attention_generic(pc, qkv, attn_only ? nullptr : w.ctx, out_attn, query_rows, n_rows, out_rows, batch, n, H, h->hd, scale, s)); }
        if (attn_only) return OCM_OK;
    } else {
        void *vt_dst = want_v ? w.vt : nullptr;
        { PROF(OCM_K_QKV, s); HIP_TRY(launch_qkv(pc, w.xn, h->ptr<char>(bp.qkv_wf), nullptr, w.q, w.k, vt_dst, out_qkv, batch, n, np, H, h->hd, want_v, s, ln1)); }
        if (out_rows && !out_attn) { PROF(OCM_K_PROBS, s); HIP_TRY(launch_attention_rows(pc, w.q, w.k, query_rows, n_rows, out_rows, batch, n, np, H, scale, s, h->hd)); }
        if (attn_only) {
            if (out_attn) {
                { PROF(OCM_K_ATTN, s); HIP_TRY(launch_attention(pc, w.q, w.k, w.vt, nullptr, w.lse, batch, n, np, H, scale, s, h->hd, w.kpart, w.kpart_bytes)); }
                { PROF(OCM_K_PROBS, s); HIP_TRY(launch_attention_probs(pc, w.q, w.k, w.lse, out_attn, batch, n, np, H, scale, s, h->hd)); }
                if (out_rows) { PROF(OCM_K_PROBS, s); HIP_TRY(launch_rows_from_probs(out_attn, query_rows, n_rows, out_rows, batch, n, H, s)); }
            }
            return OCM_OK;
        }
        { PROF(OCM_K_ATTN, s); HIP_TRY(launch_attention(pc, w.q, w.k, w.vt, w.ctx, out_attn ? w.lse : nullptr, batch, n, np, H, scale, s, h->hd, w.kpart, w.kpart_bytes)); }
        if (out_attn) {
            { PROF(OCM_K_PROBS, s); HIP_TRY(launch_attention_probs(pc, w.q, w.k, w.lse, out_attn, batch, n, np, H, scale, s, h->hd)); }
            if (out_rows) { PROF(OCM_K_PROBS, s); HIP_TRY(launch_rows_from_probs(out_attn, query_rows, n_rows, out_rows, batch, n, H, s)); }
        }
    }
    StatsOut so_mid, so_out;
    so_mid.xs = w.xn, so_mid.stats = st_mid;
    so_out.xs = w.xn, so_out.stats = st_out;
    // row centring (launch.h): each site's pairs and sums are those of x minus the row's mean at the site before it
    float *sh_in = w.shift, *sh_mid = w.shift + T;
    so_mid.shift = sh_mid, so_mid.prev_stats = st_in, so_mid.prev_shift = sh_in;
    so_out.shift = sh_in, so_out.prev_stats = st_mid, so_out.prev_shift = sh_mid;
    so_out.part = w.part;  // few rows: mlp.fc2 as split-K (launch.h)
    { PROF(OCM_K_PROJ, s); HIP_TRY(launch_linear(pc, w.ctx, h->ptr<char>(bp.proj_w), h->ptr<float>(bp.proj_b), x, x, T, D, D, OCM_EPI_BIAS_RESID_F32, s, LnFold(), so_mid)); }
    { PROF(OCM_K_FC1, s); HIP_TRY(launch_linear(pc, w.xn, h->ptr<char>(bp.fc1_wf), nullptr, nullptr, w.hid, T, h->M, D, OCM_EPI_BIAS_GELU_BF16, s, ln2)); }
    { PROF(OCM_K_FC2, s); HIP_TRY(launch_linear(pc, w.hid, h->ptr<char>(bp.fc2_w), h->ptr<float>(bp.fc2_b), x, x, T, D, h->M, OCM_EPI_BIAS_RESID_F32, s, LnFold(), so_out)); }
    return OCM_OK;
}

static int check_tiles(const ocm_vit *h, const ocm_vit_io *io, int *n_out) {
    if (!io) return fail(OCM_EINVAL, "io is null");
    if (!io->image) return fail(OCM_EINVAL, "io->image is null");
    if (!io->pos_embed) return fail(OCM_EINVAL, "io->pos_embed is null");
    if (io->batch <= 0) return fail(OCM_EINVAL, "batch %d must be positive", io->batch);
    if (io->tile_h <= 0 || io->tile_w <= 0 || io->tile_h % h->p || io->tile_w % h->p)
        return fail(OCM_EINVAL, "tile %dx%d must be a positive multiple of patch_size %d", io->tile_h, io->tile_w, h->p);
    if (io->img_stride_y % 4 || ((uintptr_t)io->image & 15) || io->img_stride_c % 4 || io->img_stride_b % 4)
        return fail(OCM_EINVAL, "image base must be 16-byte aligned and strides multiples of 4 elements");
    *n_out = (io->tile_h / h->p) * (io->tile_w / h->p) + 1;
    return OCM_OK;
}

static int run_prepare(const ocm_vit *h, const ocm_vit_io *io, float *x, int n, hipStream_t s, void *xs = nullptr,
                       float *stats = nullptr, float *shift = nullptr, float *tok_shift = nullptr) {
    PatchArgs pa{io->image, io->img_stride_b, io->img_stride_c, io->img_stride_y, io->tile_origins,
                 io->batch, io->tile_h / h->p, io->tile_w / h->p, h->p, h->C};
    if (io->patch_mask) {
        if (!h->params[h->mask_tok].set) return fail(OCM_ESTATE, "patch_mask given but parameter 'mask_token' has not been set");
        pa.mask = io->patch_mask;
        pa.mask_tok = h->ptr<float>(h->mask_tok);
    }
    StatsOut so;
    so.xs = xs, so.stats = stats;
    so.shift = shift, so.tok_shift = tok_shift;
    if (stats)  // folded first LayerNorm: the token rows also leave as split pairs with their row sums (and the rows' centring constants)
        HIP_TRY(launch_cls_rows_stats(h->ptr<float>(h->cls), io->pos_embed, x, xs, stats, io->batch, n, h->D, s, shift, tok_shift,
                                      h->ptr<float>(h->pe_b)));
    else
        HIP_TRY(launch_cls_rows(h->ptr<float>(h->cls), io->pos_embed, x, io->batch, n, h->D, s));
    PROF(OCM_K_PATCH, s);
    HIP_TRY(launch_patch_embed(h->prec, pa, h->ptr<char>(h->pe_w), h->ptr<float>(h->pe_b), io->pos_embed, x, h->D, s, so));
    return OCM_OK;
}

extern "C" int ocm_vit_prepare_tokens(ocm_vit_t *h, const ocm_vit_io *io, float *x_out) {
    if (!h || !x_out) return fail(OCM_EINVAL, "null argument");
    int rc = ocm_vit_params_ready(h);
    if (rc) return rc;
    int n = 0;
    if ((rc = check_tiles(h, io, &n))) return rc;
    return run_prepare(h, io, x_out, n, (hipStream_t)io->stream);
}

extern "C" int ocm_vit_block_forward(ocm_vit_t *h, int32_t index, float *x, int32_t batch, int32_t n_tokens,
                                     int32_t flags, float *out_attn, float *out_qkv, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    if (!h || !x) return fail(OCM_EINVAL, "null argument");
    if (index < 0 || index >= h->L) return fail(OCM_EINVAL, "block index %d out of range [0,%d)", index, h->L);
    if (batch <= 0 || n_tokens <= 1) return fail(OCM_EINVAL, "bad shape batch=%d n_tokens=%d", batch, n_tokens);
    int rc = ocm_vit_params_ready(h);
    if (rc) return rc;
    if ((rc = check_ws(h, batch, n_tokens, workspace, workspace_bytes))) return rc;
    if ((flags & OCM_OUT_ATTN) && !out_attn) return fail(OCM_EINVAL, "OCM_OUT_ATTN without out_attn");
    if ((flags & OCM_OUT_QKV) && !out_qkv) return fail(OCM_EINVAL, "OCM_OUT_QKV without out_qkv");
    const Workspace w = carve(h, batch, n_tokens, (char *)workspace);
    return run_block(h, index, w, x, batch, n_tokens, (flags & OCM_LAST_ATTN_ONLY) != 0,
                     (flags & OCM_OUT_ATTN) ? out_attn : nullptr, (flags & OCM_OUT_QKV) ? out_qkv : nullptr, nullptr, 0,
                     nullptr, (hipStream_t)stream);
}

extern "C" int ocm_vit_final_norm(ocm_vit_t *h, const float *x, float *y, int64_t rows, void *stream) {
    if (!h || !x || !y) return fail(OCM_EINVAL, "null argument");
    int rc = ocm_vit_params_ready(h);
    if (rc) return rc;
    HIP_TRY(launch_layernorm(x, h->ptr<float>(h->norm_g), h->ptr<float>(h->norm_b), y, 0, rows, h->D, h->cfg.ln_eps,
                             (hipStream_t)stream));
    return OCM_OK;
}

// The launch sequence of one forward (arguments already validated by ocm_vit_forward).
static int enqueue_forward(ocm_vit *h, const ocm_vit_io *io, int n, hipStream_t s) {
    const int fl = io->flags, B = io->batch, L = h->L;
    const bool attn_only = fl & OCM_LAST_ATTN_ONLY;
    int rc;
    const Workspace w = carve(h, B, n, (char *)io->workspace);
    const size_t T = (size_t)B * n;
    // Folding pays where it frees the N = D GEMMs from full-row tiles and removes 2 L LayerNorm launches (ViT-S/16 at
    // B = 64: -7 % per forward; one tile per call: -3 %). Forwards large enough for 512 tiles of 128 x 128 in the (T x D)
    // output never used full-row tiles and run their LayerNorm kernels at HBM speed: there the heavier epilogues cost
    // more than the launches save (4096^2 slab sweep, T = 48 k: 554 -> 561 ms), so "auto" keeps the LayerNorm kernels.
    const long t128 = (long)((T + 127) / 128) * (h->D / 128);
    const bool fold = h->folding() && (h->opt[OCM_OPT_FOLD_LN] == 2 || t128 < 512);
    if (fold) {
        if ((rc = run_prepare(h, io, w.x, n, s, w.xn, w.stats, w.shift, w.tok_shift))) return rc;
    } else if ((rc = run_prepare(h, io, w.x, n, s))) return rc;
    bool xn_ready = false;  // w.xn holds the next block's norm1(x) (fused into the previous fc2)
    for (int i = 0; i < L; ++i) {
        const int slot = i - (L - io->n_last);  // >= 0 for the returned blocks
        const bool ret = slot >= 0, last = i == L - 1;
        float *oa = (ret && (fl & OCM_OUT_ATTN)) ? io->out_attn + (size_t)slot * B * h->H * n * n : nullptr;
        float *oq = (ret && (fl & OCM_OUT_QKV)) ? io->out_qkv + (size_t)slot * 3 * B * h->H * n * h->hd : nullptr;
        float *orow = (last && (fl & OCM_OUT_ROWS)) ? io->out_rows : nullptr;
        const float *ng = last ? nullptr : h->ptr<float>(h->blk[i + 1].ln1_g), *nb = last ? nullptr : h->ptr<float>(h->blk[i + 1].ln1_b);
        if (fold) {
            if ((rc = run_block_folded(h, i, w, w.x, B, n, attn_only && last, oa, oq, io->query_rows, io->n_rows, orow, s))) return rc;
        } else if ((rc = run_block(h, i, w, w.x, B, n, attn_only && last, oa, oq, io->query_rows, io->n_rows, orow, s, xn_ready, ng, nb,
                                   &xn_ready)))
            return rc;
        if (ret && (fl & OCM_OUT_FEAT)) {
            PROF(OCM_K_LN, s);
            HIP_TRY(launch_layernorm(w.x, h->ptr<float>(h->norm_g), h->ptr<float>(h->norm_b),
                                     io->out_feat + (size_t)slot * T * h->D, 0, T, h->D, h->cfg.ln_eps, s));
        }
    }
    if (fl & OCM_OUT_TOKENS)
        HIP_TRY(hipMemcpyAsync(io->out_tokens, w.x, T * h->D * 4, hipMemcpyDeviceToDevice, s));
    if (fl & OCM_OUT_FMAP) {  // norm(x)[:, 1:] -> (B, D, hp, wp); the hidden-activation buffer is free by now
        float *yn = (float *)w.hid;
        HIP_TRY(launch_layernorm(w.x, h->ptr<float>(h->norm_g), h->ptr<float>(h->norm_b), yn, 0, T, h->D,
                                 h->cfg.ln_eps, s));
        HIP_TRY(launch_tokens_to_fmap(yn, io->out_fmap, B, n, h->D, s));
    }
    return OCM_OK;
}

extern "C" int ocm_vit_forward(ocm_vit_t *h, const ocm_vit_io *io) {
    if (!h) return fail(OCM_EINVAL, "null handle");
    int rc = ocm_vit_params_ready(h);
    if (rc) return rc;
    int n = 0;
    if ((rc = check_tiles(h, io, &n))) return rc;
    const int fl = io->flags, B = io->batch, L = h->L;
    const bool attn_only = fl & OCM_LAST_ATTN_ONLY;
    if (io->n_last < 1 || io->n_last > L) return fail(OCM_EINVAL, "n_last %d out of range [1,%d]", io->n_last, L);
    if (attn_only && (fl & (OCM_OUT_FEAT | OCM_OUT_TOKENS | OCM_OUT_QKV | OCM_OUT_FMAP)))
        return fail(OCM_EINVAL, "OCM_LAST_ATTN_ONLY excludes FEAT/TOKENS/QKV/FMAP outputs");
    if ((fl & OCM_OUT_FMAP) && !io->out_fmap) return fail(OCM_EINVAL, "OCM_OUT_FMAP without out_fmap");
    if ((fl & OCM_OUT_FMAP) && (size_t)h->M * h->esz < (size_t)h->D * 4)
        return fail(OCM_EINVAL, "OCM_OUT_FMAP needs mlp_hidden*esz >= 4*embed_dim (scratch for the normed tokens)");
    if (attn_only && io->n_last != 1) return fail(OCM_EINVAL, "OCM_LAST_ATTN_ONLY requires n_last == 1");
    if ((fl & OCM_OUT_FEAT) && !io->out_feat) return fail(OCM_EINVAL, "OCM_OUT_FEAT without out_feat");
    if ((fl & OCM_OUT_ATTN) && !io->out_attn) return fail(OCM_EINVAL, "OCM_OUT_ATTN without out_attn");
    if ((fl & OCM_OUT_QKV) && !io->out_qkv) return fail(OCM_EINVAL, "OCM_OUT_QKV without out_qkv");
    if ((fl & OCM_OUT_TOKENS) && !io->out_tokens) return fail(OCM_EINVAL, "OCM_OUT_TOKENS without out_tokens");
    if ((fl & OCM_OUT_ROWS) && (!io->out_rows || io->n_rows <= 0))
        return fail(OCM_EINVAL, "OCM_OUT_ROWS needs out_rows and n_rows > 0");
    if ((rc = check_ws(h, B, n, io->workspace, io->workspace_bytes))) return rc;
    hipStream_t s = (hipStream_t)io->stream;
    // folded-LayerNorm operands of blocks whose parameters changed: rebuilt on the caller's stream, ahead of (and never
    // inside) a captured launch sequence, which only holds their addresses
    if (h->folding() && (rc = ensure_folded(h, s))) return rc;
    if (!(fl & OCM_USE_GRAPH) || g_prof.on) return enqueue_forward(h, io, n, s);

    // ---- hipGraph path: replay when nothing the launch arguments depend on has changed --------------------
    // The legacy default stream (what torch uses unless told otherwise) cannot be captured: capture and replay on
    // a BLOCKING stream of our own instead, which the default stream orders itself with implicitly on both sides.
    if (s == nullptr) {
        if (!h->graph_stream && hipStreamCreateWithFlags(&h->graph_stream, hipStreamDefault) != hipSuccess) {
            (void)hipGetLastError();
            return enqueue_forward(h, io, n, s);
        }
        s = h->graph_stream;
    }
    ocm_vit_io key = *io;
    key.reserved = 0;
    if (h->graph_valid && memcmp(&key, &h->graph_key, sizeof key) == 0) {
        ++h->graph_hits;
        HIP_TRY(hipGraphLaunch(h->graph_exec, s));
        return OCM_OK;
    }
    // (Re-)capture the launch sequence on the caller's stream; nothing executes during capture.
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return enqueue_forward(h, io, n, s);  // stream cannot be captured (e.g. already capturing): plain launches
    }
    rc = enqueue_forward(h, io, n, s);
    const hipError_t ce = hipStreamEndCapture(s, &graph);
    if (rc || ce != hipSuccess || !graph) {  // capture refused something: nothing ran, so run it the plain way
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        h->graph_valid = false;
        return enqueue_forward(h, io, n, s);
    }
    ++h->graph_captures;
    h->graph_valid = false;
    bool have_exec = false;
    if (h->graph_exec) {  // same topology, new arguments: update the instantiated graph in place
        hipGraphNode_t err_node = nullptr;
        hipGraphExecUpdateResult res = hipGraphExecUpdateSuccess;
        if (hipGraphExecUpdate(h->graph_exec, graph, &err_node, &res) == hipSuccess && res == hipGraphExecUpdateSuccess) {
            have_exec = true;
        } else {
            (void)hipGetLastError();  // clear the sticky error of the failed update
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
    }
    if (!have_exec) {
        hipError_t ie = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            (void)hipGraphDestroy(graph);
            h->graph_exec = nullptr;
            return fail(OCM_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
        }
    }
    (void)hipGraphDestroy(graph);
    h->graph_key = key;
    h->graph_valid = true;
    HIP_TRY(hipGraphLaunch(h->graph_exec, s));
    return OCM_OK;
}

extern "C" int ocm_vit_graph_stats(const ocm_vit_t *h, uint64_t *replays, uint64_t *captures) {
    if (!h) return fail(OCM_EINVAL, "null handle");
    if (replays) *replays = h->graph_hits;
    if (captures) *captures = h->graph_captures;
    return OCM_OK;
}

// ------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------
extern "C" int ocm_op_layernorm(const float *x, const float *gamma, const float *beta, void *y, int32_t out_kind,
                                int64_t rows, int32_t dim, float eps, void *stream) {
    if (!x || !gamma || !beta || !y) return fail(OCM_EINVAL, "null argument");
    if (dim <= 0 || dim % 2 || dim > 1024) return fail(OCM_EINVAL, "dim %d must be even and <= 1024", dim);
    if (out_kind < 0 || out_kind > 2) return fail(OCM_EINVAL, "out_kind %d is not one of OCM_LN_F32 / _BF16 / _SPLIT", out_kind);
    if (out_kind == OCM_LN_SPLIT && dim % 32) return fail(OCM_EINVAL, "split-pair output needs dim %% 32 == 0 (got %d)", dim);
    HIP_TRY(launch_layernorm(x, gamma, beta, y, out_kind, rows, dim, eps, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_cast_split(const float *src, void *dst, size_t count, void *stream) {
    if (!src || !dst) return fail(OCM_EINVAL, "null argument");
    if (count % 32) return fail(OCM_EINVAL, "split-pair tensors hold multiples of 32 elements (got %zu)", count);
    HIP_TRY(launch_cast_split(src, dst, count, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_merge_split(const void *src, float *dst, size_t count, void *stream) {
    if (!src || !dst) return fail(OCM_EINVAL, "null argument");
    if (count % 32) return fail(OCM_EINVAL, "split-pair tensors hold multiples of 32 elements (got %zu)", count);
    HIP_TRY(launch_merge_split(src, dst, count, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_cast_bf16(const float *src, void *dst, size_t count, void *stream) {
    if (!src || !dst) return fail(OCM_EINVAL, "null argument");
    HIP_TRY(launch_cast_bf16(src, (bf16 *)dst, count, (hipStream_t)stream));
    return OCM_OK;
}

static int prec_of(int32_t precision, int *pc) {
    if (precision != OCM_PREC_BF16 && precision != OCM_PREC_FP32 && precision != OCM_PREC_BF16X3)
        return fail(OCM_EINVAL, "bad precision %d", precision);
    *pc = precision;
    return OCM_OK;
}

extern "C" int ocm_op_linear(int32_t precision, const void *a, const void *w, const float *bias, const float *resid,
                             void *out, int32_t M, int32_t N, int32_t K, int32_t epilogue, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!a || !w || !out) return fail(OCM_EINVAL, "null argument");
    const int kq = pc ? 32 : 64;  // elements per 128-byte operand row: 64 bf16, 32 fp32 or 32 split pairs
    if (M <= 0 || N <= 0 || N % 32 || K <= 0 || K % kq)
        return fail(OCM_EINVAL, "bad shape M=%d N=%d K=%d (N%%32, K%%%d)", M, N, K, kq);
    if (pc == 2 && (epilogue == OCM_EPI_BIAS_GELU_BF16 || epilogue == OCM_EPI_BIAS_BF16) && N % 32)
        return fail(OCM_EINVAL, "split-pair outputs need N %% 32 == 0");
    if (epilogue < 0 || epilogue > 3) return fail(OCM_EINVAL, "bad epilogue %d", epilogue);
    if (epilogue == OCM_EPI_BIAS_RESID_F32 && !resid) return fail(OCM_EINVAL, "residual epilogue without resid");
    HIP_TRY(launch_linear(pc, a, w, bias, resid, out, M, N, K, epilogue, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_linear_resid_ln_supported(int32_t D) { return linear_resid_ln_supported(D) ? 1 : 0; }

extern "C" int ocm_op_linear_resid_ln(int32_t precision, const void *a, const void *w, const float *bias,
                                      const float *resid, float *x, const float *gamma, const float *beta, void *xn,
                                      int32_t M, int32_t D, int32_t K, float eps, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!a || !w || !resid || !x || !gamma || !beta || !xn) return fail(OCM_EINVAL, "null argument");
    if (!linear_resid_ln_supported(D)) return fail(OCM_EINVAL, "row width %d not in {128, 256, 384}", D);
    if (M <= 0 || K <= 0 || K % 64) return fail(OCM_EINVAL, "bad shape M=%d K=%d (K%%64)", M, K);
    HIP_TRY(launch_linear_resid_ln(pc, a, w, bias, resid, x, gamma, beta, xn, M, D, K, eps, (hipStream_t)stream));
    return OCM_OK;
}

static int op_qkv_proj(int32_t precision, const void *a, const void *w, const float *bias, void *q, void *k, void *vt,
                       float *qkv_f32, int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    // operand copies (q, k, vt: all three or none) exist for 64-wide heads and, in split-bf16 precision, for 128-wide ones; without
    // them the projection only fills the fp32 tensor (any head width that is a multiple of 8: ocm_op_attention_generic reads it)
    const bool copies = q || k || vt;
    if (!a || !w || !bias || (copies && (!q || !k || !vt)) || (!copies && !qkv_f32)) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || n_tokens <= 0 || heads <= 0 || head_dim <= 0 || head_dim % 8) return fail(OCM_EINVAL, "bad shape");
    if (copies && head_dim != 64 && head_dim != 128)
        return fail(OCM_EINVAL, "head_dim %d: operand copies exist for 64- and 128-wide heads", head_dim);
    HIP_TRY(launch_qkv(pc, a, w, bias, q, k, vt, qkv_f32, batch, n_tokens, ocm_n_pad_for(pc, n_tokens), heads, head_dim, true,
                       (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_qkv_proj(int32_t precision, const void *a, const void *w, const float *bias, void *q, void *k,
                               void *vt, float *qkv_f32, int32_t batch, int32_t n_tokens, int32_t heads, void *stream) {
    return op_qkv_proj(precision, a, w, bias, q, k, vt, qkv_f32, batch, n_tokens, heads, 64, stream);
}

extern "C" int ocm_op_qkv_proj_hd(int32_t precision, const void *a, const void *w, const float *bias, void *q, void *k,
                                  void *vt, float *qkv_f32, int32_t batch, int32_t n_tokens, int32_t heads,
                                  int32_t head_dim, void *stream) {
    return op_qkv_proj(precision, a, w, bias, q, k, vt, qkv_f32, batch, n_tokens, heads, head_dim, stream);
}

static int op_attention(int32_t precision, const void *q, const void *k, const void *vt, void *ctx, float *lse2,
                        int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, float scale, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!q || !k || !vt || (!ctx && !lse2)) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || n_tokens <= 0 || heads <= 0) return fail(OCM_EINVAL, "bad shape");
    if (head_dim != 64 && head_dim != 128)
        return fail(OCM_EINVAL, "head_dim %d: the MFMA attention is built for 64- and 128-wide heads", head_dim);
    HIP_TRY(launch_attention(pc, q, k, vt, ctx, lse2, batch, n_tokens, ocm_n_pad_for(pc, n_tokens), heads, scale,
                             (hipStream_t)stream, head_dim));
    return OCM_OK;
}

static int op_attention_probs(int32_t precision, const void *q, const void *k, const float *lse2, float *attn, int32_t batch,
                              int32_t n_tokens, int32_t heads, int32_t head_dim, float scale, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!q || !k || !lse2 || !attn) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || n_tokens <= 0 || heads <= 0) return fail(OCM_EINVAL, "bad shape");
    if (head_dim != 64 && head_dim != 128) return fail(OCM_EINVAL, "head_dim %d not built on MFMA", head_dim);
    HIP_TRY(launch_attention_probs(pc, q, k, lse2, attn, batch, n_tokens, ocm_n_pad_for(pc, n_tokens), heads, scale,
                                   (hipStream_t)stream, head_dim));
    return OCM_OK;
}

extern "C" int ocm_op_attention(int32_t precision, const void *q, const void *k, const void *vt, void *ctx, float *lse2,
                                int32_t batch, int32_t n_tokens, int32_t heads, float scale, void *stream) {
    return op_attention(precision, q, k, vt, ctx, lse2, batch, n_tokens, heads, 64, scale, stream);
}

extern "C" int ocm_op_attention_hd(int32_t precision, const void *q, const void *k, const void *vt, void *ctx, float *lse2,
                                   int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, float scale,
                                   void *stream) {
    return op_attention(precision, q, k, vt, ctx, lse2, batch, n_tokens, heads, head_dim, scale, stream);
}

extern "C" int ocm_op_attention_generic(int32_t precision, const float *qkv_f32, void *ctx, float *attn, int32_t batch,
                                        int32_t n_tokens, int32_t heads, int32_t head_dim, float scale, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!qkv_f32 || (!ctx && !attn)) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || n_tokens <= 0 || heads <= 0 || head_dim <= 0) return fail(OCM_EINVAL, "bad shape");
    if (head_dim % 4 || head_dim > 512 || n_tokens > 8192 || (pc == 2 && (heads * head_dim) % 32))
        return fail(OCM_EINVAL, "generic attention: head_dim %d must be a multiple of 4 up to 512, at most 8192 tokens, and "
                                "heads * head_dim a multiple of 32 for a split-bf16 context", head_dim);
    HIP_TRY(launch_attention_generic(pc, qkv_f32, ctx, attn, nullptr, 0, nullptr, batch, n_tokens, heads, head_dim, scale,
                                     (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_attention_probs(int32_t precision, const void *q, const void *k, const float *lse2, float *attn,
                                      int32_t batch, int32_t n_tokens, int32_t heads, float scale, void *stream) {
    return op_attention_probs(precision, q, k, lse2, attn, batch, n_tokens, heads, 64, scale, stream);
}

extern "C" int ocm_op_attention_probs_hd(int32_t precision, const void *q, const void *k, const float *lse2, float *attn,
                                         int32_t batch, int32_t n_tokens, int32_t heads, int32_t head_dim, float scale,
                                         void *stream) {
    return op_attention_probs(precision, q, k, lse2, attn, batch, n_tokens, heads, head_dim, scale, stream);
}

extern "C" int ocm_op_attention_rows(int32_t precision, const void *q, const void *k, const int32_t *query_rows,
                                     int32_t n_rows, float *rows, int32_t batch, int32_t n_tokens, int32_t heads,
                                     float scale, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!q || !k || !rows) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || n_tokens <= 1 || heads <= 0 || n_rows <= 0) return fail(OCM_EINVAL, "bad shape");
    HIP_TRY(launch_attention_rows(pc, q, k, query_rows, n_rows, rows, batch, n_tokens, ocm_n_pad_for(pc, n_tokens), heads, scale,
                                  (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_attention_map(const float *attn, float *maps, int32_t b, int32_t heads, int32_t n_tokens,
                                    int32_t query, int32_t hf, int32_t wf, int32_t p, void *stream) {
    if (!attn || !maps) return fail(OCM_EINVAL, "null argument");
    if (hf * wf + 1 != n_tokens) return fail(OCM_EINVAL, "hf*wf+1 = %d != n_tokens %d", hf * wf + 1, n_tokens);
    if (query < 0 || query >= n_tokens) return fail(OCM_EINVAL, "query %d out of range", query);
    HIP_TRY(launch_attention_map(attn, maps, b, heads, n_tokens, query, hf, wf, p, (hipStream_t)stream));
    return OCM_OK;
}

// ------------------------------------------------------------------------------------------
// sliding-window post-processing
// ------------------------------------------------------------------------------------------
extern "C" int ocm_op_tile_postprocess(const float *rows, float *maps, int32_t tiles, int32_t heads, int32_t n_rows,
                                       int32_t pixels, void *stream) {
    if (!rows || !maps) return fail(OCM_EINVAL, "null argument");
    if (tiles <= 0 || heads <= 0 || n_rows <= 0 || pixels <= 0) return fail(OCM_EINVAL, "bad shape");
    HIP_TRY(launch_tile_postprocess(rows, maps, tiles, heads, n_rows, pixels, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_bilinear_upsample(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w,
                                        int32_t scale, void *stream) {
    if (!src || !dst) return fail(OCM_EINVAL, "null argument");
    if (tiles <= 0 || h <= 0 || w <= 0 || scale <= 0) return fail(OCM_EINVAL, "bad shape");
    HIP_TRY(launch_bilinear_up(src, dst, tiles, h, w, scale, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_nearest_upsample(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t rep,
                                       void *stream) {
    if (!src || !dst) return fail(OCM_EINVAL, "null argument");
    if (tiles <= 0 || h <= 0 || w <= 0 || rep <= 0) return fail(OCM_EINVAL, "bad shape");
    HIP_TRY(launch_nearest_up(src, dst, tiles, h, w, rep, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_stitch(const float *crops, float *out, const double *ramp, int32_t n, int32_t window,
                             int32_t stride, void *stream) {
    if (!crops || !out || !ramp) return fail(OCM_EINVAL, "null argument");
    if (n <= 0 || stride <= 0 || window <= stride || window > 3 * stride)
        return fail(OCM_EINVAL, "stitch needs stride < window <= 3*stride (got window %d stride %d)", window, stride);
    HIP_TRY(launch_stitch(crops, out, ramp, n, window, stride, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_normalize_u8(const float *img, int64_t count, void *scratch, uint8_t *out, uint64_t *hist256,
                                   void *stream) {
    if (!img || !scratch || !out || !hist256) return fail(OCM_EINVAL, "null argument");
    if (count <= 0) return fail(OCM_EINVAL, "bad count");
    HIP_TRY(launch_normalize_u8(img, (size_t)count, (float *)scratch, out, (unsigned long long *)hist256,
                                (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_head_mean(const float *rows, float *maps, int32_t tiles, int32_t heads, int32_t n_rows,
                                int32_t pixels, void *stream) {
    if (!rows || !maps) return fail(OCM_EINVAL, "null argument");
    if (tiles <= 0 || heads <= 0 || n_rows <= 0 || pixels <= 0) return fail(OCM_EINVAL, "bad shape");
    HIP_TRY(launch_head_mean(rows, maps, tiles, heads, n_rows, pixels, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_image_to_gray_u8(const float *image, int64_t stride_c, int32_t chans, int64_t count, uint8_t *out,
                                       uint64_t *hist256, void *stream) {
    if (!image || !out) return fail(OCM_EINVAL, "null argument");
    if (count <= 0 || (chans != 1 && chans != 3)) return fail(OCM_EINVAL, "image_to_gray_u8 takes 1 or 3 planes");
    HIP_TRY(launch_image_to_gray_u8(image, stride_c, chans, (size_t)count, out, (unsigned long long *)hist256,
                                    (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_blend_u8(const uint8_t *img, const uint8_t *att, int64_t count, double alpha,
                               double one_minus_alpha, uint8_t *out, uint64_t *hist256, void *stream) {
    if (!img || !att || !out) return fail(OCM_EINVAL, "null argument");
    if (count <= 0) return fail(OCM_EINVAL, "bad count");
    HIP_TRY(launch_blend_u8(img, att, (size_t)count, alpha, one_minus_alpha, out, (unsigned long long *)hist256,
                            (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_pixel_shuffle(const float *lin, float *out, int32_t batch, int32_t hp, int32_t wp, int32_t c_out,
                                    int32_t sh, void *stream) {
    if (!lin || !out) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || hp <= 0 || wp <= 0 || c_out <= 0 || sh <= 0) return fail(OCM_EINVAL, "bad shape");
    HIP_TRY(launch_pixel_shuffle(lin, out, batch, hp, wp, c_out, sh, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_stitch_image_u8(const float *image, int64_t stride_c, int64_t stride_y, int32_t chans, int32_t height,
                                      int32_t width, uint8_t *out, const double *ramp, int32_t n, int32_t window,
                                      int32_t stride, uint64_t *hist256, void *stream) {
    if (!image || !out || !ramp) return fail(OCM_EINVAL, "null argument");
    if (chans != 1 && chans != 3) return fail(OCM_EINVAL, "stitch_image_u8 takes 1 or 3 planes");
    if (n <= 0 || stride <= 0 || window <= stride || window > 3 * stride || height <= 0 || width <= 0)
        return fail(OCM_EINVAL, "stitch needs stride < window <= 3*stride (got window %d stride %d)", window, stride);
    HIP_TRY(launch_stitch_image_u8(image, stride_c, stride_y, chans, height, width, out, ramp, n, window, stride,
                                   (unsigned long long *)hist256, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_weighted_u8(const float *heat, const uint8_t *img, int64_t count, void *scratch, uint8_t *result,
                                  uint8_t *att_u8, uint64_t *hist_result, uint64_t *hist_att, void *stream) {
    if (!heat || !img || !scratch || !result || !att_u8 || !hist_result || !hist_att) return fail(OCM_EINVAL, "null argument");
    if (count <= 0) return fail(OCM_EINVAL, "bad count");
    HIP_TRY(launch_weighted_u8(heat, img, (size_t)count, (float *)scratch, result, att_u8, (unsigned long long *)hist_result,
                               (unsigned long long *)hist_att, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_histogram_u8(const uint8_t *img, int64_t count, uint64_t *hist256, void *stream) {
    if (!img || !hist256) return fail(OCM_EINVAL, "null argument");
    if (count <= 0) return fail(OCM_EINVAL, "bad count");
    HIP_TRY(launch_histogram_u8(img, (size_t)count, (unsigned long long *)hist256, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_median_filter(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t size,
                                    void *stream) {
    if (!src || !dst || src == dst) return fail(OCM_EINVAL, "null or aliased argument");
    if (tiles <= 0 || h <= 0 || w <= 0 || size < 1 || size > 15) return fail(OCM_EINVAL, "bad shape / size (1..15)");
    HIP_TRY(launch_median_filter(src, dst, tiles, h, w, size, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_downscale_centre(const float *src, float *dst, int32_t tiles, int32_t h, int32_t w, int32_t factor,
                                       void *stream) {
    if (!src || !dst) return fail(OCM_EINVAL, "null argument");
    if (tiles <= 0 || h <= 0 || w <= 0 || factor < 1 || h % factor || w % factor)
        return fail(OCM_EINVAL, "bad shape: %dx%d is not a multiple of the factor %d", h, w, factor);
    HIP_TRY(launch_downscale_centre(src, dst, tiles, h, w, factor, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_im2col3x3(int32_t precision, const float *in, void *out, int32_t batch, int32_t h, int32_t w,
                                int32_t channels, int32_t relu, void *stream) {
    int pc = 0, rc = prec_of(precision, &pc);
    if (rc) return rc;
    if (!in || !out) return fail(OCM_EINVAL, "null argument");
    if (batch <= 0 || h <= 0 || w <= 0 || channels <= 0 || channels % 32)
        return fail(OCM_EINVAL, "bad shape batch=%d h=%d w=%d channels=%d (channels %% 32)", batch, h, w, channels);
    HIP_TRY(launch_im2col3x3(pc, in, out, batch, h, w, channels, relu, (hipStream_t)stream));
    return OCM_OK;
}

extern "C" int ocm_op_threshold_u8(const uint8_t *img, uint8_t *mask, int64_t count, int32_t thresh, void *stream) {
    if (!img || !mask) return fail(OCM_EINVAL, "null argument");
    HIP_TRY(launch_threshold_u8(img, mask, (size_t)count, thresh, (hipStream_t)stream));
    return OCM_OK;
}

// Otsu threshold of a 256-bin histogram (host). Restates cv2.threshold(..., THRESH_OTSU)'s
// getThreshVal_Otsu_8u (OpenCV 4.6.0 modules/imgproc/src/thresh.cpp — an un-vendored dependency of the
// reference, opencv-python 4.6.0.66; parity unpinned: cv2 is not installable here).
extern "C" int32_t ocm_otsu_threshold(const uint64_t *hist256, int64_t count) {
    if (!hist256 || count <= 0) return -OCM_EINVAL;
    const double scale = 1.0 / (double)count;
    double mu = 0;
    for (int i = 0; i < 256; ++i) mu += i * (double)hist256[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0;
    int max_val = 0;
    for (int i = 0; i < 256; ++i) {
        const double p_i = (double)hist256[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        const double lo = q1 < q2 ? q1 : q2, hi = q1 < q2 ? q2 : q1;
        if (lo < 1.1920928955078125e-07 || hi > 1.0 - 1.1920928955078125e-07) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    return max_val;
}

// ------------------------------------------------------------------------------------------
// sliding-window index math (integers, host) — sw_processing.py:151-163 and SURVEY §8-e
// ------------------------------------------------------------------------------------------
extern "C" int32_t ocm_sw_count(int32_t size, int32_t stride) {
    if (stride <= 0) return 0;
    const int32_t stop = size - 2 * stride;  // range(0, size - 2*stride, stride)
    return stop <= 0 ? 0 : (stop + stride - 1) / stride;
}

extern "C" int32_t ocm_sw_origins(int32_t height, int32_t width, int32_t stride, int32_t *origins_yx, int32_t cap) {
    if (!origins_yx || stride <= 0) return -OCM_EINVAL;
    // NB the reference unpacks `height, width = image.size` from a PIL (width, height) pair and
    // loops y over the first, x over the second (sw_processing.py:153-157); for the square images
    // it is used on the two agree. Here y walks `height`, x walks `width`.
    const int32_t ny = ocm_sw_count(height, stride), nx = ocm_sw_count(width, stride);
    if ((int64_t)ny * nx > cap) return -OCM_ENOMEM;
    int32_t n = 0;
    for (int32_t iy = 0; iy < ny; ++iy)
        for (int32_t ix = 0; ix < nx; ++ix) {
            origins_yx[2 * n] = iy * stride;
            origins_yx[2 * n + 1] = ix * stride;
            ++n;
        }
    return n;
}

extern "C" int32_t ocm_sw_shard(int32_t n_tiles, int32_t world, int32_t rank, int32_t *begin, int32_t *end) {
    if (world <= 0 || rank < 0 || rank >= world || n_tiles < 0 || !begin || !end) return -OCM_EINVAL;
    const int32_t share = (n_tiles + world - 1) / world;
    int32_t b = rank * share, e = b + share;
    if (b > n_tiles) b = n_tiles;
    if (e > n_tiles) e = n_tiles;
    *begin = b;
    *end = e;
    return share;
}
