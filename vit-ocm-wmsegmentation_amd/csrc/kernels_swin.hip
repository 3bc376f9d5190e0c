// kernels_swin.hip — the Swin-specific stages of the Swin-T forward (BASELINE config 5;
// arithmetic = transformers/models/swin/modeling_swin.py, the reference's dependency at
// Allen_data_Backbone/train.py:70-85). The projections and the MLP reuse the MFMA GEMMs of kernels_gemm.hip;
// here are the pieces with Swin's own data movement:
//   swin_embed_kernel      4x4 stride-4 conv + bias + LayerNorm            (SwinEmbeddings.forward :219-244)
//   swin_ln_kernel         LayerNorm with an output row stride (K padded to the GEMM step) and, for patch
//                          merging, the 2x2 neighbour gather fused in front   (SwinPatchMerging.forward :309-326)
//   swin_bias_perm_kernel  relative-position bias table -> dense bias in the attention kernel's register order
//   swin_wattn_kernel      (shifted-)window attention: cyclic shift, window partition, q k^T * scale + bias + mask,
//                          softmax, P v, window reverse, shift back — one wavefront per (window, head), K and V^T
//                          of the window staged in LDS, both contractions on MFMA   (SwinLayer.forward :529-574)
//   swin_wattn_f32_kernel  the same in plain fp32 FMAs (OCM_PREC_FP32)
//   swin_pool_head_kernel  final LayerNorm + mean over tokens + classifier       (SwinModel :887-892, :1052)
#include "launch.h"

// ------------------------------------------------------------------------------------------
// patch embedding + LayerNorm (SwinEmbeddings :273-291: Conv2d(k = 4, stride 4), flatten, LayerNorm) on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: fp32 products, so every precision mode starts from the same fp32 residual stream).
// A wavefront owns 32 tokens: y^T (C0 channels x 32 tokens) = W (C0 x K) . patch^T (K x 32), K = 16 * channels. The lane's
// token is its accumulator COLUMN, so the 4 x 4 patch is twelve (or four) 16-byte loads per lane, coalesced over the 32
// neighbouring patches of an image row, and the LayerNorm statistics of a token are a sum over the lane's registers plus one
// lane <-> lane + 32 exchange. W stays in registers across a grid-stride loop over token tiles.
// ------------------------------------------------------------------------------------------
template <int CH, int CF>  // input channels (1 or 3), C0 / 32
__global__ __launch_bounds__(256) void swin_embed_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                         const float *__restrict__ bias, const float *__restrict__ g,
                                                         const float *__restrict__ be, float *__restrict__ x, int B, int S,
                                                         float eps) {
    constexpr int K = CH * 16, KK = K / 2, C0 = CF * 32;
    __shared__ __attribute__((aligned(16))) float par[3 * C0];  // bias | gamma | beta
    for (int i = threadIdx.x; i < C0; i += 256) {
        par[i] = bias[i];
        par[C0 + i] = g[i];
        par[2 * C0 + i] = be[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    const int hp = S / 4;
    const size_t T = (size_t)B * hp * hp, tiles = (T + 31) / 32;
    float wa[CF][KK];  // A operand of k step kk: W[32 mf + r][2 kk + h]
#pragma unroll
    for (int mf = 0; mf < CF; ++mf)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wa[mf][kk] = w[(32 * mf + r) * K + 2 * kk + h];
    for (size_t tile = gw; tile < tiles; tile += nw) {
        const size_t t = tile * 32 + r, tc = t < T ? t : T - 1;
        const int b = (int)(tc / ((size_t)hp * hp)), pi = (int)(tc - (size_t)b * hp * hp);
        const int py = pi / hp, px = pi - py * hp;
        const float *p = img + (((size_t)b * CH) * S + py * 4) * S + px * 4;
        f32x4 pv[CH][4];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) pv[ch][dy] = *(const f32x4 *)(p + ((size_t)ch * S + dy) * S);
        f32x16 acc[CF];
#pragma unroll
        for (int mf = 0; mf < CF; ++mf)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mf][e] = 0.f;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dp = 0; dp < 2; ++dp) {  // k = 16 ch + 4 dy + 2 dp + h
                    const float bv = h ? pv[ch][dy][2 * dp + 1] : pv[ch][dy][2 * dp];
#pragma unroll
                    for (int mf = 0; mf < CF; ++mf) acc[mf] = mfma32f(wa[mf][ch * 8 + dy * 2 + dp], bv, acc[mf]);
                }
        // register 4 q + e of fragment mf = channel 32 mf + 8 q + 4 h + e of the lane's token
        float sum = 0.f;
#pragma unroll
        for (int mf = 0; mf < CF; ++mf)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *(const f32x4 *)(par + 32 * mf + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[mf][4 * q + e] += bb[e];
                    sum += acc[mf][4 * q + e];
                }
            }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C0);
        float var = 0.f;
#pragma unroll
        for (int mf = 0; mf < CF; ++mf)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = acc[mf][e] - mean;
                var = fmaf(d, d, var);
            }
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * (1.0f / C0) + eps);
        if (t < T) {
            float *xo = x + t * C0;
#pragma unroll
            for (int mf = 0; mf < CF; ++mf)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 32 * mf + 8 * q + 4 * h;
                    const f32x4 gg = *(const f32x4 *)(par + C0 + c), bb = *(const f32x4 *)(par + 2 * C0 + c);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (acc[mf][4 * q + e] - mean) * rstd * gg[e] + bb[e];
                    *(f32x4 *)(xo + c) = o;
                }
        }
    }
}

hipError_t launch_swin_embed(const float *img, const float *w, const float *bias, const float *g, const float *be,
                             float *x, int batch, int chans, int size, int c0, float eps, hipStream_t s) {
    if ((chans != 1 && chans != 3) || c0 % 32 || c0 <= 0 || c0 > 128 || size % 4) return hipErrorInvalidValue;
    const size_t T = (size_t)batch * (size / 4) * (size / 4), tiles = (T + 31) / 32;
    const unsigned blocks = (unsigned)((tiles + 3) / 4 < 1536 ? (tiles + 3) / 4 : 1536);  // up to six workgroups per CU
#define OCM_EMBED(CH, CF)                                                                                               \
    swin_embed_kernel<CH, CF><<<dim3(blocks), dim3(256), 0, s>>>(img, w, bias, g, be, x, batch, size, eps)
    switch ((chans == 3 ? 4 : 0) + c0 / 32 - 1) {
        case 0: OCM_EMBED(1, 1); break;
        case 1: OCM_EMBED(1, 2); break;
        case 2: OCM_EMBED(1, 3); break;
        case 3: OCM_EMBED(1, 4); break;
        case 4: OCM_EMBED(3, 1); break;
        case 5: OCM_EMBED(3, 2); break;
        case 6: OCM_EMBED(3, 3); break;
        default: OCM_EMBED(3, 4); break;
    }
#undef OCM_EMBED
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LayerNorm -> GEMM operand rows of stride ldy (columns [dim, ldy) zero). MERGE: row = merged token,
// its 4*C inputs are the four 2x2 neighbours in the order (row0,col0), (row1,col0), (row0,col1), (row1,col1).
// ------------------------------------------------------------------------------------------
template <class E>
__device__ __forceinline__ void store4(E *row, int c, const f32x4 &v);  // columns c .. c+3 of an operand row
template <>
__device__ __forceinline__ void store4<float>(float *row, int c, const f32x4 &v) { *(f32x4 *)(row + c) = v; }
template <>
__device__ __forceinline__ void store4<bf16>(bf16 *row, int c, const f32x4 &v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *(bf16x4 *)(row + c) = o;
}
template <>
__device__ __forceinline__ void store4<sp32>(sp32 *row, int c, const f32x4 &v) {  // [32 x hi | 32 x lo] groups (common.h)
    bf16x4 hi, lo;
    split4(v, hi, lo);
    char *p = (char *)row + sp_off(c);
    *(bf16x4 *)p = hi;
    *(bf16x4 *)(p + 64) = lo;
}

// LPR lanes per row (32: two rows per wavefront for dim <= 128; 64 otherwise), V float4 chunks per lane.
template <class E, bool MERGE, int LPR, int V>
__global__ __launch_bounds__(256) void swin_ln_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                      const float *__restrict__ be, E *__restrict__ y, size_t rows,
                                                      int dim, int ldy, float eps, int Hin, int Win) {
    constexpr int RPW = 64 / LPR;  // rows per wavefront
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const size_t row = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    const size_t rr = live ? row : rows - 1;  // clamped: every lane takes part in the shuffles
    const float *src[4];
    int C = dim;
    if (MERGE) {
        C = dim >> 2;
        const int Ho = (Hin + 1) >> 1, Wo = (Win + 1) >> 1;  // SwinPatchMerging.maybe_pad: an odd side gets a row / column of zeros
        const size_t b = rr / ((size_t)Ho * Wo);
        const int rem = (int)(rr - b * Ho * Wo), yo = rem / Wo, xo = rem - yo * Wo;
#pragma unroll
        for (int sgm = 0; sgm < 4; ++sgm) {
            const int r = sgm & 1, c = sgm >> 1;
            const bool inside = 2 * yo + r < Hin && 2 * xo + c < Win;
            src[sgm] = inside ? x + ((b * Hin + 2 * yo + r) * Win + 2 * xo + c) * (size_t)C : nullptr;
        }
    } else {
        src[0] = x + rr * (size_t)dim;
    }
    f32x4 v[V];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (sub + LPR * i) * 4;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c < dim) {
            if (MERGE) {
                if (src[c / C]) t = *(const f32x4 *)(src[c / C] + c % C);
            } else {
                t = *(const f32x4 *)(src[0] + c);
            }
        }
        v[i] = t;
        sum += (t[0] + t[1]) + (t[2] + t[3]);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)dim;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if ((sub + LPR * i) * 4 < dim) {
            const f32x4 d = v[i] - mean;
            var += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = rsqrtf(var / (float)dim + eps);
    if (!live) return;
    E *dst = y + row * (size_t)ldy;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (sub + LPR * i) * 4;
        if (c < dim) {
            const f32x4 gg = *(const f32x4 *)(g + c), bb = *(const f32x4 *)(be + c);
            store4<E>(dst, c, (v[i] - mean) * rstd * gg + bb);
        } else if (c < ldy) {
            store4<E>(dst, c, f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    for (int c = (sub + LPR * V) * 4; c < ldy; c += LPR * 4) store4<E>(dst, c, f32x4{0.f, 0.f, 0.f, 0.f});
}

template <class E, bool MERGE>
static hipError_t launch_swin_ln_e(const float *x, const float *g, const float *be, E *y, size_t rows, int dim, int ldy,
                                   float eps, int Hin, int Win, hipStream_t s) {
    if (dim % 4 || ldy % 4 || (MERGE && (dim / 4) % 4)) return hipErrorInvalidValue;
    const dim3 block(256);
#define OCM_LN_CASE(lpr, v)                                                                                         \
    if (dim <= lpr * v * 4) {                                                                                       \
        const dim3 grid((unsigned)((rows + 4 * (64 / lpr) - 1) / (4 * (64 / lpr))));                                 \
        swin_ln_kernel<E, MERGE, lpr, v><<<grid, block, 0, s>>>(x, g, be, y, rows, dim, ldy, eps, Hin, Win);        \
        return hipGetLastError();                                                                                   \
    }
    OCM_LN_CASE(32, 1) OCM_LN_CASE(64, 1) OCM_LN_CASE(64, 2) OCM_LN_CASE(64, 3) OCM_LN_CASE(64, 6)
#undef OCM_LN_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_swin_ln(int prec, const float *x, const float *g, const float *be, void *y, size_t rows, int dim,
                          int ldy, float eps, bool merge, int Hin, int Win, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    if (prec == 2) {
        if (ldy % 32) return hipErrorInvalidValue;
        return merge ? launch_swin_ln_e<sp32, true>(x, g, be, (sp32 *)y, rows, dim, ldy, eps, Hin, Win, s)
                     : launch_swin_ln_e<sp32, false>(x, g, be, (sp32 *)y, rows, dim, ldy, eps, Hin, Win, s);
    }
    if (prec) {
        return merge ? launch_swin_ln_e<float, true>(x, g, be, (float *)y, rows, dim, ldy, eps, Hin, Win, s)
                     : launch_swin_ln_e<float, false>(x, g, be, (float *)y, rows, dim, ldy, eps, Hin, Win, s);
    }
    return merge ? launch_swin_ln_e<bf16, true>(x, g, be, (bf16 *)y, rows, dim, ldy, eps, Hin, Win, s)
                 : launch_swin_ln_e<bf16, false>(x, g, be, (bf16 *)y, rows, dim, ldy, eps, Hin, Win, s);
}

// ------------------------------------------------------------------------------------------
// SwinLayer.maybe_pad (modeling_swin.py SwinLayer.forward): a grid that is not a multiple of the window is padded with ZERO rows
// (after layernorm_before) to the right and at the bottom, the attention half runs on the padded grid and the padded
// positions' outputs are dropped. Two byte-moving kernels around the unchanged projections and window-attention kernels:
//   swin_pad_rows_kernel : operand rows (B, H, W, row) -> (B, Hp, Wp, row), zeros at the padded positions;
//   swin_crop_add_kernel : x (B, H, W, C) += y (B, Hp, Wp, C) at the real positions (the residual of the half).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swin_pad_rows_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int H, int W,
                                                            int Hp, int Wp, int chunks, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / chunks;
        const int c = (int)(i - row * chunks);
        const size_t b = row / ((size_t)Hp * Wp);
        const int rem = (int)(row - b * Hp * Wp), y = rem / Wp, xx = rem - y * Wp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y < H && xx < W) v = src[((b * H + y) * W + xx) * (size_t)chunks + c];
        dst[i] = v;
    }
}

__global__ __launch_bounds__(256) void swin_crop_add_kernel(f32x4 *__restrict__ x, const f32x4 *__restrict__ yp, int H, int W, int Hp,
                                                            int Wp, int chunks, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / chunks;
        const int c = (int)(i - row * chunks);
        const size_t b = row / ((size_t)H * W);
        const int rem = (int)(row - b * H * W), y = rem / W, xx = rem - y * W;
        x[i] += yp[((b * Hp + y) * Wp + xx) * (size_t)chunks + c];
    }
}

// rows of `row_bytes` bytes (a multiple of 16)
hipError_t launch_swin_pad_rows(const void *src, void *dst, int batch, int H, int W, int Hp, int Wp, size_t row_bytes,
                                hipStream_t s) {
    if (row_bytes % 16 || Hp < H || Wp < W || batch <= 0) return hipErrorInvalidValue;
    const int chunks = (int)(row_bytes / 16);
    const size_t total = (size_t)batch * Hp * Wp * chunks;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    swin_pad_rows_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const f32x4 *)src, (f32x4 *)dst, H, W, Hp, Wp, chunks, total);
    return hipGetLastError();
}

hipError_t launch_swin_crop_add(float *x, const float *yp, int batch, int H, int W, int Hp, int Wp, int C, hipStream_t s) {
    if (C % 4 || Hp < H || Wp < W || batch <= 0) return hipErrorInvalidValue;
    const int chunks = C / 4;
    const size_t total = (size_t)batch * H * W * chunks;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    swin_crop_add_kernel<<<dim3(blocks), dim3(256), 0, s>>>((f32x4 *)x, (const f32x4 *)yp, H, W, Hp, Wp, chunks, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// relative-position bias (SwinRelativePositionBias :329-370) in the attention kernels' order, times log2(e):
//   out[head][qt][c][lane][e4] = table[index(i, j)][head] * log2e,  c = sub*4 + e/4, e4 = e%4 (16 accumulator
//   registers e per key half `sub`), i = qt*32 + (lane & 31), j = sub*32 + key_of_reg(e, lane >> 5): one 16-B load
//   per lane and chunk c, 1 KiB contiguous per wavefront. Keys j >= ws*ws get -1e30 (the padding mask), queries
//   i >= ws*ws 0.
// `dense` (optional) receives the plain [head][ws*ws][ws*ws] table (natural-log domain) for the fp32 kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swin_bias_perm_kernel(const float *__restrict__ table, float *__restrict__ perm,
                                                             float *__restrict__ dense, int heads, int ws) {
    const int A = ws * ws;
    const int total = heads * 2 * 64 * 32;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int e4 = idx & 3, lane = (idx >> 2) & 63, c8 = (idx >> 8) & 7, qt = (idx >> 11) & 1, head = idx >> 12;
        const int sub = c8 >> 2, e = (c8 & 3) * 4 + e4;
        const int i = qt * 32 + (lane & 31), j = sub * 32 + key_of_reg(e, lane >> 5);
        float v = 0.f;
        if (j >= A) {
            v = -1e30f;
        } else if (i < A) {
            const int yi = i / ws, xi = i - yi * ws, yj = j / ws, xj = j - yj * ws;
            v = table[((yi - yj + ws - 1) * (2 * ws - 1) + (xi - xj + ws - 1)) * heads + head] * 1.4426950408889634f;
        }
        perm[idx] = v;
    }
    if (dense) {
        const int tot2 = heads * A * A;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < tot2; idx += gridDim.x * 256) {
            const int j = idx % A, i = (idx / A) % A, head = idx / (A * A);
            const int yi = i / ws, xi = i - yi * ws, yj = j / ws, xj = j - yj * ws;
            dense[idx] = table[((yi - yj + ws - 1) * (2 * ws - 1) + (xi - xj + ws - 1)) * heads + head];
        }
    }
}

hipError_t launch_swin_bias_perm(const float *table, float *perm, float *dense, int heads, int ws, hipStream_t s) {
    swin_bias_perm_kernel<<<dim3(64), dim3(256), 0, s>>>(table, perm, dense, heads, ws);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// window attention
// ------------------------------------------------------------------------------------------
struct WinGeom {
    int H, W, ws, shift, nWx, nW, heads;
};
// token (row of the (B, H*W, C) stream) of position p of window (b, wy, wx): the window tiles the image rolled
// by -shift (cyclic_shift :617-626 + window_partition :486-495), so its source pixel is (+shift) mod size; the
// output goes back to the same token (window_reverse + reverse roll).
// (`ws` is passed separately: a compile-time 7 in the Swin-T instantiation turns the divisions into multiplies)
__device__ __forceinline__ size_t win_token(const WinGeom &g, int ws, int b, int wy, int wx, int p) {
    const int py = p / ws, px = p - py * ws;
    int y = wy * ws + py + g.shift, x = wx * ws + px + g.shift;
    if (y >= g.H) y -= g.H;
    if (x >= g.W) x -= g.W;
    return ((size_t)b * g.H + y) * g.W + x;
}
// region id of get_attn_mask (:584-607) for position p of the window, in the SHIFTED frame
__device__ __forceinline__ int win_region(const WinGeom &g, int ws, int wy, int wx, int p) {
    const int py = p / ws, px = p - py * ws;
    const int ys = wy * ws + py, xs = wx * ws + px;
    const int ry = (ys >= g.H - ws) + (ys >= g.H - g.shift), rx = (xs >= g.W - ws) + (xs >= g.W - g.shift);
    return ry * 3 + rx;
}

// Eight consecutive rows of one column of a row-major 16-bit LDS tile as an MFMA fragment: two ds_read_b64_tr_b16 (gfx950),
// the second four rows (256 bytes at 64-byte rows) further down. `p` is this lane's address for the first block.
__device__ __forceinline__ bf16x8 tr_read8(const char *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * 64));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
#else
    return bf16x8{};
#endif
}

// bf16 / MFMA: one wavefront per (b, window, head), head_dim 32, ws*ws <= 64 positions.
// S^T = K.Q^T with the K rows in pi order (common.h): registers hold keys, the lane holds the query, so the
// row max / sum are in-register reductions plus one lane <-> lane+32 exchange and the exponentiated tile is
// directly the B operand of O^T += V^T.P^T (same scheme as kernels_attn.hip).
template <int WS>  // window side known at compile time (7 for Swin-T), or 0 = read it from the geometry
__global__ __launch_bounds__(256) void swin_wattn_kernel(const bf16 *__restrict__ qkv, int ld, bf16 *__restrict__ ctx,
                                                         int ldc, const float *__restrict__ bias_perm, WinGeom g,
                                                         int total, float scale2) {
    __shared__ __attribute__((aligned(16))) char smem[4 * (64 * 64 + 32 * 128 + 64)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int id = blockIdx.x * 4 + wave;
    if (id >= total) return;  // no workgroup barrier below: every wave owns its LDS slice
    char *Ks = smem + wave * (64 * 64 + 32 * 128 + 64);  // K: 64 keys x 64 B
    char *Vs = Ks + 64 * 64;                             // V: [64 keys][32 dims] row-major, 64-byte rows (read transposed)
    unsigned char *Rg = (unsigned char *)(Vs + 32 * 128);
    const int head = id % g.heads, wlin = (id / g.heads) % g.nW, b = id / (g.heads * g.nW);
    const int wy = wlin / g.nWx, wx = wlin - wy * g.nWx;
    const int ws = WS ? WS : g.ws;
    const int A = ws * ws, C = g.heads * 32;
    const bf16 *base = qkv + head * 32;

    // All global loads of the wave are issued in one burst (K / V rows of the window's tokens, the Q fragments of
    // both query tiles), then K rows and V^T columns go to LDS (padding keys: exact zeros).
    bf16x8 kreg[4], vreg[4], qf[2][2];
    size_t qtok[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i, key = idx >> 2, ch = idx & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) kreg[i][e] = vreg[i][e] = (bf16)0.f;
        if (key < A) {
            const bf16 *row = base + win_token(g, ws, b, wy, wx, key) * (size_t)ld + ch * 8;
            kreg[i] = *(const bf16x8 *)(row + C);
            vreg[i] = *(const bf16x8 *)(row + 2 * C);
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        qtok[qt] = win_token(g, ws, b, wy, wx, min(qt * 32 + r, A - 1));
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[qt][s] = *(const bf16x8 *)(base + qtok[qt] * (size_t)ld + 16 * s + 8 * h);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i, key = idx >> 2, ch = idx & 3;
        *(bf16x8 *)(Ks + key * 64 + ((ch ^ ((key >> 2) & 3)) << 4)) = kreg[i];  // chunk swizzle: conflict-free b128 reads
        *(bf16x8 *)(Vs + key * 64 + ch * 16) = vreg[i];  // V row-major [64 keys][32 dims]: read transposed below (tr_read8)
    }
    const bool masked = g.shift > 0 && (wy == g.H / ws - 1 || wx == g.nWx - 1);  // wave-uniform
    if (masked) Rg[lane] = (unsigned char)(lane < A ? win_region(g, ws, wy, wx, lane) : 0);
    const int pr = pi_row(r);
    const float *bp = bias_perm + (size_t)head * 2 * 64 * 32 + lane * 4;
    f32x4 bvq[2][8];  // the relative-position bias of both query tiles, requested with the K / V / Q burst
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) bvq[qt][c8] = *(const f32x4 *)(bp + qt * 64 * 32 + c8 * 256);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (qt * 32 >= A) break;
        const int qi = qt * 32 + r;
        f32x16 S[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 a = *(const bf16x8 *)(Ks + (sub * 32 + pr) * 64 + (((2 * s + h) ^ ((pr >> 2) & 3)) << 4));
                S[sub] = mfma32(a, qf[qt][s], S[sub]);
            }
        }
        float mx = -INFINITY;
        const int myreg = masked ? Rg[min(qi, A - 1)] : 0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 bv = bvq[qt][sub * 4 + e4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(S[sub][e4 * 4 + e], scale2, bv[e]);
                    if (masked) {
                        const int j = sub * 32 + key_of_reg(e4 * 4 + e, h);
                        if (j < A && Rg[j] != myreg) v += -100.0f * 1.4426950408889634f;
                    }
                    S[sub][e4 * 4 + e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(S[sub][e] - mx);
                S[sub][e] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);
        f32x16 O;
#pragma unroll
        for (int e = 0; e < 16; ++e) O[e] = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = (bf16)S[sub][8 * s2 + e];
                const bf16x8 a = tr_read8(Vs + (sub * 32 + 16 * s2 + 8 * h + ((lane >> 2) & 3)) * 64 +
                                          (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
                O = mfma32(a, pb, O);
            }
        // Lane (r, h) holds dims {8g + 4h + e}: trade half of them with lane (r, h ^ 1) so that each lane owns 16
        // consecutive dims and the two lanes of a query write one contiguous 64-byte row segment.
        {
            const float inv = 1.0f / l;
            uint32_t pk[8];  // pk[2g], pk[2g+1]: dims 8g + 4h + {0,1}, {2,3} as packed bf16
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    bf16x2 t;
                    t[0] = (bf16)(O[4 * gq + 2 * e] * inv);
                    t[1] = (bf16)(O[4 * gq + 2 * e + 1] * inv);
                    pk[2 * gq + e] = __builtin_bit_cast(uint32_t, t);
                }
            uint32_t out[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // h = 0 sends groups 2, 3 and keeps 0, 1; h = 1 the other way round
                const uint32_t send = h ? pk[k] : pk[4 + k];
                const uint32_t recv = (uint32_t)__shfl_xor((int)send, 32, 64);
                const uint32_t keep = h ? pk[4 + k] : pk[k];
                // h = 0: dims [0..3]=keep g0, [4..7]=recv g0, [8..11]=keep g1, [12..15]=recv g1
                // h = 1: dims [16..19]=recv g2, [20..23]=keep g2, [24..27]=recv g3, [28..31]=keep g3
                const int gsel = k >> 1, w = k & 1;
                out[4 * gsel + (h ? 2 : 0) + w] = keep;
                out[4 * gsel + (h ? 0 : 2) + w] = recv;
            }
            if (qi < A) {
                bf16 *dst = ctx + qtok[qt] * (size_t)ldc + head * 32 + 16 * h;
                *(uint4 *)dst = uint4{out[0], out[1], out[2], out[3]};
                *(uint4 *)(dst + 8) = uint4{out[4], out[5], out[6], out[7]};
            }
        }
    }
}

// split-bf16 (OCM_PREC_BF16X3): swin_wattn_kernel on [hi | lo] pairs. One head of one token is exactly one 128-byte
// group of the operand row (head_dim 32: 32 x hi | 32 x lo), so q / k / v fragments are 16-byte pieces of it; scores and
// context are three MFMAs per product (mfma32x3), the probabilities are split in registers, the context leaves as pairs.
// LDS per wavefront: K 64 keys x 128 B (hi | lo), V as two row-major [64 keys][64 B] images (hi, lo) read transposed.
constexpr int WATTN_X3_WAVES = 3;  // 16.1 KiB of LDS per wavefront: three workgroups of three per CU
template <int WS>
__global__ __launch_bounds__(WATTN_X3_WAVES * 64) void swin_wattn_x3_kernel(const char *__restrict__ qkv, int ld, char *__restrict__ ctx,
                                                            int ldc, const float *__restrict__ bias_perm, WinGeom g,
                                                            int total, float scale2) {
    constexpr int PER_WAVE = 64 * 128 + 2 * 32 * 128 + 64;
    __shared__ __attribute__((aligned(16))) char smem[WATTN_X3_WAVES * PER_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int id = blockIdx.x * WATTN_X3_WAVES + wave;
    if (id >= total) return;  // no workgroup barrier below: every wave owns its LDS slice
    char *Ks = smem + wave * PER_WAVE;  // K: 64 keys x 128 B, chunks 0..3 = hi, 4..7 = lo (lds_off swizzle)
    char *Vh = Ks + 64 * 128;           // V hi: [64 keys][32 dims] row-major (64-byte rows); V lo follows
    char *Vl = Vh + 32 * 128;
    unsigned char *Rg = (unsigned char *)(Vl + 32 * 128);
    const int head = id % g.heads, wlin = (id / g.heads) % g.nW, b = id / (g.heads * g.nW);
    const int wy = wlin / g.nWx, wx = wlin - wy * g.nWx;
    const int ws = WS ? WS : g.ws;
    const int A = ws * ws, C = g.heads * 32;
    const size_t ldb = (size_t)ld * 4, ldcb = (size_t)ldc * 4;  // row strides in bytes (4 bytes per element)
    const char *base = qkv + head * 128;

    // one burst of global loads: the K / V groups of the window's tokens (8 chunks of 16 B per group: lane -> (key, chunk))
    // and the Q fragments of both query tiles
    bf16x8 kreg[8], vreg[8], qh[2][2], ql[2][2];
    size_t qtok[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i, key = idx >> 3, ch = idx & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) kreg[i][e] = vreg[i][e] = (bf16)0.f;
        if (key < A) {
            const char *row = base + win_token(g, ws, b, wy, wx, key) * ldb + ch * 16;
            kreg[i] = *(const bf16x8 *)(row + (size_t)C * 4);
            vreg[i] = *(const bf16x8 *)(row + (size_t)2 * C * 4);
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        qtok[qt] = win_token(g, ws, b, wy, wx, min(qt * 32 + r, A - 1));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const char *p = base + qtok[qt] * ldb + (16 * s + 8 * h) * 2;
            qh[qt][s] = *(const bf16x8 *)p;
            ql[qt][s] = *(const bf16x8 *)(p + 64);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i, key = idx >> 3, ch = idx & 7;
        *(bf16x8 *)(Ks + lds_off(key, ch)) = kreg[i];
        // V stays row-major ([64 keys][32 dims] per half, 64-byte rows): the context product reads it TRANSPOSED from LDS
        // (ds_read_b64_tr_b16 below) instead of scattering sixteen-bit elements into a V^T image here
        *(bf16x8 *)((ch < 4 ? Vh : Vl) + key * 64 + (ch & 3) * 16) = vreg[i];
    }
    const bool masked = g.shift > 0 && (wy == g.H / ws - 1 || wx == g.nWx - 1);  // wave-uniform
    if (masked) Rg[lane] = (unsigned char)(lane < A ? win_region(g, ws, wy, wx, lane) : 0);
    const int pr = pi_row(r);
    const float *bp = bias_perm + (size_t)head * 2 * 64 * 32 + lane * 4;
    // the relative-position bias of both query tiles, requested with the K / V / Q burst (one L2 round trip per tile was exposed
    // between the score MFMAs and the softmax before)
    f32x4 bvq[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) bvq[qt][c8] = *(const f32x4 *)(bp + qt * 64 * 32 + c8 * 256);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (qt * 32 >= A) break;
        const int qi = qt * 32 + r;
        f32x16 S[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 kh = *(const bf16x8 *)(Ks + lds_off(sub * 32 + pr, 2 * s + h));
                const bf16x8 kl = *(const bf16x8 *)(Ks + lds_off(sub * 32 + pr, 4 + 2 * s + h));
                S[sub] = mfma32x3(kh, kl, qh[qt][s], ql[qt][s], S[sub]);
            }
        }
        float mx = -INFINITY;
        const int myreg = masked ? Rg[min(qi, A - 1)] : 0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 bv = bvq[qt][sub * 4 + e4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(S[sub][e4 * 4 + e], scale2, bv[e]);
                    if (masked) {
                        const int j = sub * 32 + key_of_reg(e4 * 4 + e, h);
                        if (j < A && Rg[j] != myreg) v += -100.0f * 1.4426950408889634f;
                    }
                    S[sub][e4 * 4 + e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(S[sub][e] - mx);
                S[sub][e] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);
        f32x16 O;
#pragma unroll
        for (int e = 0; e < 16; ++e) O[e] = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = S[sub][8 * s2 + e];
                    const bf16 t = (bf16)pv;
                    ph[e] = t;
                    pl[e] = (bf16)(pv - (float)t);
                }
                // A operand V^T[d = lane & 31][8 keys from sub * 32 + 16 s2 + 8 h]: two transposed reads of 4 keys x 16 dims
                // per 16-lane group (lane 4 q + p of a group addresses key row q, dims 4 p .. 4 p + 3 of the block; lane i
                // receives dim i of the four keys). Every lane is active here (whole wavefronts leave the kernel together).
                const int voff = (sub * 32 + 16 * s2 + 8 * h + ((lane >> 2) & 3)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
                const bf16x8 vh = tr_read8(Vh + voff), vl = tr_read8(Vl + voff);
                O = mfma32x3(vh, vl, ph, pl, O);
            }
        // Lane (r, h) holds dims {8g + 4h + e} in fp32: the pair halves go out as four 8-byte pieces per half
        if (qi < A) {
            const float inv = 1.0f / l;
            char *dst = ctx + qtok[qt] * ldcb + head * 128;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = O[4 * gq + e] * inv;
                bf16x4 oh, ol;
                split4(o, oh, ol);
                char *p = dst + (8 * gq + 4 * h) * 2;
                *(bf16x4 *)p = oh;
                *(bf16x4 *)(p + 64) = ol;
            }
        }
    }
}

// fp32 (OCM_PREC_FP32): one wavefront per (b, window, head); lane = query, K / V rows broadcast from LDS.
__global__ __launch_bounds__(256) void swin_wattn_f32_kernel(const float *__restrict__ qkv, int ld,
                                                             float *__restrict__ ctx, int ldc,
                                                             const float *__restrict__ bias_dense, WinGeom g, int total,
                                                             float scale) {
    __shared__ float smem[4 * (2 * 49 * 32)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int id = blockIdx.x * 4 + wave;
    if (id >= total) return;
    float *Ks = smem + wave * (2 * 49 * 32), *Vs = Ks + 49 * 32;
    const int head = id % g.heads, wlin = (id / g.heads) % g.nW, b = id / (g.heads * g.nW);
    const int wy = wlin / g.nWx, wx = wlin - wy * g.nWx;
    const int ws = g.ws;
    const int A = ws * ws, C = g.heads * 32;
    const float *base = qkv + head * 32;
    for (int idx = lane; idx < A * 8; idx += 64) {
        const int key = idx >> 3, ch = idx & 7;
        const float *row = base + win_token(g, ws, b, wy, wx, key) * (size_t)ld + ch * 4;
        *(f32x4 *)(Ks + key * 32 + ch * 4) = *(const f32x4 *)(row + C);
        *(f32x4 *)(Vs + key * 32 + ch * 4) = *(const f32x4 *)(row + 2 * C);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    if (lane >= A) return;
    const size_t qtok = win_token(g, ws, b, wy, wx, lane);
    float q[32];
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
        const f32x4 t = *(const f32x4 *)(base + qtok * (size_t)ld + d);
        q[d] = t[0]; q[d + 1] = t[1]; q[d + 2] = t[2]; q[d + 3] = t[3];
    }
    const bool masked = g.shift > 0;
    const int myreg = masked ? win_region(g, ws, wy, wx, lane) : 0;
    const float *brow = bias_dense + ((size_t)head * A + lane) * A;
    float sc[49];
    float mx = -INFINITY;
#pragma unroll 7
    for (int j = 0; j < 49; ++j) {
        float s = -INFINITY;
        if (j < A) {
            float d0 = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) d0 = fmaf(q[d], Ks[j * 32 + d], d0);
            s = d0 * scale + brow[j];
            if (masked && win_region(g, ws, wy, wx, j) != myreg) s += -100.0f;
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    float o[32], l = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll 7
    for (int j = 0; j < 49; ++j) {
        if (j < A) {
            const float p = __expf(sc[j] - mx);
            l += p;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] = fmaf(p, Vs[j * 32 + d], o[d]);
        }
    }
    const float inv = 1.0f / l;
    float *dst = ctx + qtok * (size_t)ldc + head * 32;
#pragma unroll
    for (int d = 0; d < 32; d += 4) *(f32x4 *)(dst + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
}

hipError_t launch_swin_window_attention(int prec, const void *qkv, int ld, void *ctx, int ldc, const float *bias_perm,
                                        const float *bias_dense, int batch, int H, int W, int ws, int shift, int heads,
                                        hipStream_t s) {
    if (ws * ws > 49 + 15 || H % ws || W % ws) return hipErrorInvalidValue;
    WinGeom g{H, W, ws, shift, W / ws, (H / ws) * (W / ws), heads};
    const long total = (long)batch * g.nW * heads;
    if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((total + 3) / 4);
    const float scale = 0.17677669529663687f;  // 32^-0.5 (SwinAttention.scaling :408)
    if (prec == 2) {
        if (ld % 32 || ldc % 32) return hipErrorInvalidValue;
        const dim3 gx3((unsigned)((total + WATTN_X3_WAVES - 1) / WATTN_X3_WAVES)), bx3(WATTN_X3_WAVES * 64);
        if (ws == 7)
            swin_wattn_x3_kernel<7><<<gx3, bx3, 0, s>>>((const char *)qkv, ld, (char *)ctx, ldc, bias_perm, g,
                                                                        (int)total, scale * 1.4426950408889634f);
        else
            swin_wattn_x3_kernel<0><<<gx3, bx3, 0, s>>>((const char *)qkv, ld, (char *)ctx, ldc, bias_perm, g,
                                                                        (int)total, scale * 1.4426950408889634f);
    } else if (prec)
        swin_wattn_f32_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const float *)qkv, ld, (float *)ctx, ldc, bias_dense, g,
                                                                  (int)total, scale);
    else if (ws == 7)
        swin_wattn_kernel<7><<<dim3(blocks), dim3(256), 0, s>>>((const bf16 *)qkv, ld, (bf16 *)ctx, ldc, bias_perm, g,
                                                                 (int)total, scale * 1.4426950408889634f);
    else
        swin_wattn_kernel<0><<<dim3(blocks), dim3(256), 0, s>>>((const bf16 *)qkv, ld, (bf16 *)ctx, ldc, bias_perm, g,
                                                                 (int)total, scale * 1.4426950408889634f);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// MLP half of a SwinLayer in one kernel (split-bf16, narrow stages: C = 96 or 128)
//     x += fc2(gelu(fc1(LayerNorm(x))))            modeling_swin.py SwinLayer.forward :668-672, SwinIntermediate / SwinOutput
// At 8e5 token rows and C = 96 the unfused chain (LayerNorm, fc1 + GELU, fc2 + residual) streams 4 GB per layer, most of
// it the 4C-wide hidden activations (4 bytes per element as split pairs): it runs at the HBM rate, not the matrix rate.
// Here a wavefront owns 32 tokens and keeps everything that belongs to them on chip:
//   * LayerNorm of its rows in registers, split into pairs: the B operand x^T of the first product (2 * C / 16 fragments);
//   * hidden^T (32 hidden units x 32 tokens per step) = W1 rows . x^T with the W1 rows in pi order (common.h), so that
//     after bias + GELU the accumulator registers ARE the B operand fragments of the second product (the scheme of the
//     attention kernels, kernels_attn.hip): y^T (C x 32 tokens) += W2[:, step] . hidden^T;
//   * W1 / W2 pass through a three-stage `buffer_load ... lds` ring shared by the eight wavefronts of the workgroup (24 KiB
//     per step of 32 hidden units at C = 96), one counted vmcnt wait and one barrier per step;
//   * epilogue: + b2 + x, fp32, in place. HBM traffic: x once in, once out.
// ------------------------------------------------------------------------------------------
template <int CG, int HG, int NW, int NSTAGE>  // C / 32, hidden / 32, wavefronts per workgroup, LDS stages of the W ring
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 3) void swin_mlp_x3_kernel(float *__restrict__ x, const float *__restrict__ gam,
                                                             const float *__restrict__ bet, const char *__restrict__ w1,
                                                             const float *__restrict__ b1, const char *__restrict__ w2,
                                                             const float *__restrict__ b2, int T, float eps) {
    constexpr int C = CG * 32, HID = HG * 32, W1B = CG * 4096, STAGE = W1B + C * 128;
    constexpr int PIECES = STAGE / 1024, PPW = PIECES / NW, NS = 2 * CG;  // 1-KiB DMA pieces per step / per wave; k slices of 16
    static_assert(PIECES % NW == 0, "the wavefronts share the pieces of a step evenly");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "one or two steps in flight");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float *b1s = (float *)(smem + NSTAGE * STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tok0 = (blockIdx.x * NW + wave) * 32;
    const bool live = tok0 + r < T;
    const size_t tok = (size_t)min(tok0 + r, T - 1);

    for (int i = tid; i < HID; i += NW * 64) b1s[i] = b1[i];

    // piece pc = jj * NW + wave of a step: pieces [0, 4 CG) = W1 (image pc >> 2 = k group, rows (pc & 3) * 8 .. + 7 of the
    // step's 32 hidden units), the rest = W2 (rows (pc - 4 CG) * 8 .. + 7 of the C outputs, the step's k group)
    int voff[PPW];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int pc = jj * NW + wave;
            if (pc < 4 * CG) {
                const int rho = (pc & 3) * 8 + lrow;
                voff[jj] = rho * (C * 4) + (pc >> 2) * 128 + ((slot ^ ((rho >> 1) & 7)) << 4);
            } else {
                const int c = (pc - 4 * CG) * 8 + lrow;
                voff[jj] = c * (HID * 4) + ((slot ^ ((c >> 1) & 7)) << 4);
            }
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)w1, 0, (unsigned)(HID * C * 4), 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)w2, 0, (unsigned)(HID * C * 4), 0x00020000);
#define OCM_MLP_DMA(j, st)                                                                                               \
    do {                                                                                                                 \
        _Pragma("unroll") for (int jj = 0; jj < PPW; ++jj) {                                                             \
            const int pc = jj * NW + wave;                                                                               \
            char *dst_ = smem + (st) * STAGE + pc * 1024;                                                                \
            if (pc < 4 * CG)                                                                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)dst_, 16, voff[jj], (j) * (32 * C * 4), 0, 0);    \
            else                                                                                                         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)dst_, 16, voff[jj], (j) * 128, 0, 0);             \
        }                                                                                                                \
    } while (0)
#else
#define OCM_MLP_DMA(j, st) (void)0
#endif
    OCM_MLP_DMA(0, 0);
    if (NSTAGE == 3 && HG > 1) OCM_MLP_DMA(1, 1);

    // LayerNorm of the wave's rows: lane (r, h) holds channels 16 s + 8 h .. + 7 of token r for every slice s
    bf16x8 xh[NS], xl[NS];
    {
        const float *xr = x + tok * C + 8 * h;
        f32x4 v[NS][2];
        float sum = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            v[sI][0] = *(const f32x4 *)(xr + 16 * sI);
            v[sI][1] = *(const f32x4 *)(xr + 16 * sI + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += v[sI][0][e] + v[sI][1][e];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float var = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = v[sI][0][e] - mean, d1 = v[sI][1][e] - mean;
                var = fmaf(d0, d0, fmaf(d1, d1, var));
            }
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * (1.0f / C) + eps);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const f32x4 g0 = *(const f32x4 *)(gam + 16 * sI + 8 * h), g1 = *(const f32x4 *)(gam + 16 * sI + 8 * h + 4);
            const f32x4 e0 = *(const f32x4 *)(bet + 16 * sI + 8 * h), e1 = *(const f32x4 *)(bet + 16 * sI + 8 * h + 4);
            split8((v[sI][0] - mean) * rstd * g0 + e0, (v[sI][1] - mean) * rstd * g1 + e1, xh[sI], xl[sI]);
        }
    }
    // rows, parameters and the first two steps have landed (the builtin, so that hipcc's own bookkeeping sees it and does
    // not wait again inside the loop with a count that covers the step in flight)
    __builtin_amdgcn_s_waitcnt(0x0F70);

    f32x16 Y[CG];
#pragma unroll
    for (int mf = 0; mf < CG; ++mf)
#pragma unroll
        for (int e = 0; e < 16; ++e) Y[mf][e] = 0.f;
    const int pr = pi_row(r);
    int sc = 0, si = NSTAGE - 1;
    for (int j = 0; j < HG; ++j) {
        if (NSTAGE == 3 && j + 1 < HG)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");  // the younger step may still be on its way
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + NSTAGE - 1 < HG) OCM_MLP_DMA(j + NSTAGE - 1, si);  // into the stage of step j - 1: everybody is past it
        const char *W1s = smem + sc * STAGE, *W2s = W1s + W1B;
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const char *img = W1s + (sI >> 1) * 4096;
            const bf16x8 ah = *(const bf16x8 *)(img + lds_off(pr, (sI & 1) * 2 + h));
            const bf16x8 al = *(const bf16x8 *)(img + lds_off(pr, 4 + (sI & 1) * 2 + h));
            S = mfma32x3(ah, al, xh[sI], xl[sI], S);
        }
        // register e of lane half h holds hidden unit 32 j + key_of_reg(e, h): registers 0..7 and 8..15 are runs of eight
        bf16x8 ph[2], pl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float *bp = b1s + 32 * j + 16 * s2 + 8 * h;
            const f32x4 c0 = *(const f32x4 *)bp, c1 = *(const f32x4 *)(bp + 4);
            f32x4 u0, u1;
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const f32x2 a = gelu_erf2(f32x2{S[8 * s2 + e] + c0[e], S[8 * s2 + e + 1] + c0[e + 1]});
                const f32x2 b = gelu_erf2(f32x2{S[8 * s2 + 4 + e] + c1[e], S[8 * s2 + 4 + e + 1] + c1[e + 1]});
                u0[e] = a[0]; u0[e + 1] = a[1];
                u1[e] = b[0]; u1[e + 1] = b[1];
            }
            split8(u0, u1, ph[s2], pl[s2]);
        }
#pragma unroll
        for (int mf = 0; mf < CG; ++mf)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 vh = *(const bf16x8 *)(W2s + lds_off(32 * mf + r, 2 * s2 + h));
                const bf16x8 vl = *(const bf16x8 *)(W2s + lds_off(32 * mf + r, 4 + 2 * s2 + h));
                Y[mf] = mfma32x3(vh, vl, ph[s2], pl[s2], Y[mf]);
            }
        sc = sc == NSTAGE - 1 ? 0 : sc + 1;
        si = si == NSTAGE - 1 ? 0 : si + 1;
    }
#undef OCM_MLP_DMA
    if (!live) return;
    // y^T: lane (r, h) register 4 g + e of fragment mf = channel 32 mf + 8 g + 4 h + e of token r
    float *xo = x + tok * C;
#pragma unroll
    for (int mf = 0; mf < CG; ++mf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * mf + 8 * g + 4 * h;
            const f32x4 old = *(const f32x4 *)(xo + c), bb = *(const f32x4 *)(b2 + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Y[mf][4 * g + e] + bb[e] + old[e];
            *(f32x4 *)(xo + c) = o;
        }
}

// layernorm_before + the fused q | k | v projection of a narrow-stage SwinLayer in one kernel (split-bf16, C = 96 / 128):
//     qkv (T, 3C) pairs = LayerNorm(x) Wqkv^T + b                                    modeling_swin.py SwinLayer.forward :641, SwinSelfAttention :430-432
// The first half of swin_mlp_x3_kernel: a wavefront owns 32 tokens, normalises their rows in registers into the B operand
// x^T, and walks the 3C / 32 chunks of 32 output features — W rows in pi order through a three-stage LDS-DMA ring — whose
// accumulator registers (eight consecutive features per lane and run) leave as 16-byte pieces of the tokens' pair rows.
// HBM traffic: x once in, qkv once out (1.23 GB at 8e5 rows, against 0.6 + 1.23 GB for LayerNorm kernel + GEMM, the
// latter at 2.9 TB/s on three-K-step tiles).
// (GELU: the same kernel as layernorm_after + SwinIntermediate — N = 4C output features, exact-erf GELU before the split — for
// the stage whose MLP half does not fit swin_mlp_x3_kernel's registers: C = 192)
template <int CG, int NW, bool GELU>
__global__ __launch_bounds__(NW * 64, CG <= 4 ? 4 : 2) void swin_lnqkv_x3_kernel(const float *__restrict__ x, const float *__restrict__ gam,
                                                                  const float *__restrict__ bet, const char *__restrict__ w,
                                                                  const float *__restrict__ bias, char *__restrict__ qkv, int T,
                                                                  int N, float eps) {
    constexpr int C = CG * 32, STAGE = CG * 4096, NSTAGE = 3, PIECES = 4 * CG, PPW = PIECES / NW, NS = 2 * CG;
    const int NF = N >> 5;  // chunks of 32 output features
    static_assert(PIECES % NW == 0, "the wavefronts share the pieces of a step evenly");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float *bs = (float *)(smem + NSTAGE * STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tok0 = (blockIdx.x * NW + wave) * 32;
    const bool live = tok0 + r < T;
    const size_t tok = (size_t)min(tok0 + r, T - 1);
    for (int i = tid; i < N; i += NW * 64) bs[i] = bias[i];
    int voff[PPW];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int pc = jj * NW + wave, rho = (pc & 3) * 8 + lrow;
            voff[jj] = rho * (C * 4) + (pc >> 2) * 128 + ((slot ^ ((rho >> 1) & 7)) << 4);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, (unsigned)(N * C * 4), 0x00020000);
#define OCM_QKV_DMA(j, st)                                                                                              \
    do {                                                                                                                \
        _Pragma("unroll") for (int jj = 0; jj < PPW; ++jj)                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(smem + (st) * STAGE + (jj * NW + wave) * 1024), 16, \
                                                     voff[jj], (j) * (32 * C * 4), 0, 0);                               \
    } while (0)
#else
#define OCM_QKV_DMA(j, st) (void)0
#endif
    OCM_QKV_DMA(0, 0);
    if (NF > 1) OCM_QKV_DMA(1, 1);
    bf16x8 xh[NS], xl[NS];
    {
        const float *xr = x + tok * C + 8 * h;
        f32x4 v[NS][2];
        float sum = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            v[sI][0] = *(const f32x4 *)(xr + 16 * sI);
            v[sI][1] = *(const f32x4 *)(xr + 16 * sI + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += v[sI][0][e] + v[sI][1][e];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float var = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = v[sI][0][e] - mean, d1 = v[sI][1][e] - mean;
                var = fmaf(d0, d0, fmaf(d1, d1, var));
            }
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * (1.0f / C) + eps);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const f32x4 g0 = *(const f32x4 *)(gam + 16 * sI + 8 * h), g1 = *(const f32x4 *)(gam + 16 * sI + 8 * h + 4);
            const f32x4 e0 = *(const f32x4 *)(bet + 16 * sI + 8 * h), e1 = *(const f32x4 *)(bet + 16 * sI + 8 * h + 4);
            split8((v[sI][0] - mean) * rstd * g0 + e0, (v[sI][1] - mean) * rstd * g1 + e1, xh[sI], xl[sI]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const int pr = pi_row(r);
    char *orow = qkv + tok * ((size_t)N * 4);
    int sc = 0, si = 2;
    for (int j = 0; j < NF; ++j) {
        if (j + 1 < NF)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 2 < NF) OCM_QKV_DMA(j + 2, si);
        const char *Ws = smem + sc * STAGE;
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const char *img = Ws + (sI >> 1) * 4096;
            const bf16x8 ah = *(const bf16x8 *)(img + lds_off(pr, (sI & 1) * 2 + h));
            const bf16x8 al = *(const bf16x8 *)(img + lds_off(pr, 4 + (sI & 1) * 2 + h));
            S = mfma32x3(ah, al, xh[sI], xl[sI], S);
        }
        // register e of lane half h = feature 32 j + key_of_reg(e, h): two runs of eight -> two 16-byte pieces per half
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float *bp = bs + 32 * j + 16 * s2 + 8 * h;
            const f32x4 c0 = *(const f32x4 *)bp, c1 = *(const f32x4 *)(bp + 4);
            f32x4 u0, u1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u0[e] = S[8 * s2 + e] + c0[e];
                u1[e] = S[8 * s2 + 4 + e] + c1[e];
            }
            if (GELU) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 a = gelu_erf2(f32x2{u0[e], u0[e + 1]}), b = gelu_erf2(f32x2{u1[e], u1[e + 1]});
                    u0[e] = a[0]; u0[e + 1] = a[1];
                    u1[e] = b[0]; u1[e + 1] = b[1];
                }
            }
            bf16x8 ph, pl;
            split8(u0, u1, ph, pl);
            if (live) {
                char *p = orow + (size_t)j * 128 + (16 * s2 + 8 * h) * 2;
                *(bf16x8 *)p = ph;
                *(bf16x8 *)(p + 64) = pl;
            }
        }
        sc = sc == NSTAGE - 1 ? 0 : sc + 1;
        si = si == NSTAGE - 1 ? 0 : si + 1;
    }
#undef OCM_QKV_DMA
}

// C = 192 (stage 1 of Swin-T: 96 registers of x fragments, a 72 KiB ring, two workgroups per CU) is the widest that fits
bool swin_lnqkv_fused_supported(int prec, int C) { return prec == 2 && (C == 96 || C == 128 || C == 192); }

// LayerNorm(x) W^T + b [-> GELU] as split pairs, W (N, C): the q | k | v projection (N = 3C) or SwinIntermediate (N = 4C, gelu)
hipError_t launch_swin_lnlinear(int prec, const float *x, const float *g, const float *be, const void *w, const float *bias,
                                void *out, size_t T, int C, int N, bool gelu, float eps, hipStream_t s) {
    if (!swin_lnqkv_fused_supported(prec, C) || T == 0 || T > 0x7fffffffu || N <= 0 || N % 32 || N > 4096) return hipErrorInvalidValue;
    constexpr int NW = 4;
    const dim3 grid((unsigned)((T + NW * 32 - 1) / (NW * 32))), block(NW * 64);
    const int lds = 3 * (C / 32 * 4096) + N * 4;
    const void *kern = nullptr;
#define OCM_LNL(CG_, G_)                                                                                                \
    do {                                                                                                                \
        kern = (const void *)swin_lnqkv_x3_kernel<CG_, NW, G_>;                                                         \
        if (lds > 64 * 1024) {                                                                                          \
            static unsigned long long optin = 0;                                                                        \
            int dev = 0;                                                                                                \
            if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;                                           \
            if (!(optin >> (dev & 63) & 1)) {                                                                           \
                if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);    \
                    e != hipSuccess)                                                                                    \
                    return e;                                                                                           \
                optin |= 1ull << (dev & 63);                                                                            \
            }                                                                                                           \
        }                                                                                                               \
        swin_lnqkv_x3_kernel<CG_, NW, G_><<<grid, block, lds, s>>>(x, g, be, (const char *)w, bias, (char *)out, (int)T, N, \
                                                                   eps);                                                \
    } while (0)
    if (C == 96) {
        if (gelu) OCM_LNL(3, true); else OCM_LNL(3, false);
    } else if (C == 128) {
        if (gelu) OCM_LNL(4, true); else OCM_LNL(4, false);
    } else {
        if (gelu) OCM_LNL(6, true); else OCM_LNL(6, false);
    }
#undef OCM_LNL
    return hipGetLastError();
}

hipError_t launch_swin_lnqkv(int prec, const float *x, const float *g, const float *be, const void *w, const float *bias,
                             void *qkv, size_t T, int C, float eps, hipStream_t s) {
    return launch_swin_lnlinear(prec, x, g, be, w, bias, qkv, T, C, 3 * C, false, eps, s);
}

bool swin_mlp_fused_supported(int prec, int C, int hidden) { return prec == 2 && hidden == 4 * C && (C == 96 || C == 128); }

hipError_t launch_swin_mlp(int prec, float *x, const float *g, const float *be, const void *w1, const float *b1,
                           const void *w2, const float *b2, size_t T, int C, int hidden, float eps, hipStream_t s) {
    if (!swin_mlp_fused_supported(prec, C, hidden) || T == 0 || T > 0x7fffffffu) return hipErrorInvalidValue;
    // C = 96: four wavefronts on a two-stage ring, 50 KiB and 162 registers -> three workgroups per CU;
    // C = 128: 237 registers -> eight wavefronts, one workgroup per CU, three stages
    constexpr int NW96 = 4, NS96 = 2, NW128 = 8, NS128 = 3;
    const int nw = C == 96 ? NW96 : NW128, nst = C == 96 ? NS96 : NS128;
    const dim3 grid((unsigned)((T + nw * 32 - 1) / (nw * 32))), block(nw * 64);
    const int lds = nst * (C / 32 * 4096 + C * 128) + hidden * 4;
    static unsigned long long optin[2] = {0, 0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const void *kern = C == 96 ? (const void *)swin_mlp_x3_kernel<3, 12, NW96, NS96>
                               : (const void *)swin_mlp_x3_kernel<4, 16, NW128, NS128>;
    unsigned long long &mask = optin[C == 96 ? 0 : 1];
    if (!(mask >> (dev & 63) & 1)) {
        if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); e != hipSuccess) return e;
        mask |= 1ull << (dev & 63);
    }
    if (C == 96)
        swin_mlp_x3_kernel<3, 12, NW96, NS96><<<grid, block, lds, s>>>(x, g, be, (const char *)w1, b1, (const char *)w2, b2,
                                                                        (int)T, eps);
    else
        swin_mlp_x3_kernel<4, 16, NW128, NS128><<<grid, block, lds, s>>>(x, g, be, (const char *)w1, b1, (const char *)w2, b2,
                                                                          (int)T, eps);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Attention half of a SwinLayer in one kernel (split-bf16; C = 96, stage 0 of Swin-T: all of it; C = 192, stage 1: up to the
// context, o_proj stays a GEMM)
//     x += o_proj(window_attention(q | k | v of LayerNorm(x)))      modeling_swin.py SwinLayer.forward :641-666
// (layernorm_before :641, SwinSelfAttention :430-470 with the relative-position bias :329-370 and the shift mask :584-607,
// SwinSelfOutput :476-484, the residual :664-666). Unfused, this half streams x, the q | k | v pairs (three times the size
// of x), the context pairs and x again through HBM — 12 x the bytes of x per layer at 8e5 token rows; here a token's row is
// read twice and written once, everything else stays on chip:
//   * a workgroup = four wavefronts = TWO windows; wavefront (win, qt) owns window positions 32 qt .. 32 qt + 31 (49 real
//     ones at ws = 7): LayerNorm of its rows in registers as the B operand x^T (swin_lnqkv_x3_kernel's scheme);
//   * per head four steps on one weight chunk each (32 rows of Wq, Wk, Wv — pi order — or the head's 128-byte column group
//     of all C rows of Wo; 4 KiB x C / 32 each) through a three-stage `buffer_load ... lds` ring shared by the four
//     wavefronts, one counted vmcnt wait and one barrier per step:
//       q: accumulator registers + bias, split -> ARE the B fragments of the score product (never leave registers);
//       k, v: + bias, split -> the window's K image (64 keys x 128 B) and row-major V images (hi, lo) in LDS, padding keys 0;
//       o: scores / softmax / context exactly as swin_wattn_x3_kernel (same operands, same order: the context pairs have
//          the same bits as the unfused path's), V^T read from LDS with its rows in pi order, so that the normalised context
//          registers ARE the B fragments of y^T (C x 32 tokens) += Wo[:, head] . ctx^T;
//   * epilogue: + bias + x, fp32, in place (a window's tokens belong to no other window: no other wavefront reads them).
// FUSE_PROJ false (C = 192: the x fragments take 96 registers, a second set of 96 for y^T does not fit beside them): three
// chunks per head, the fourth step only waits for the K / V images; the context leaves as pairs (swin_wattn_x3_kernel's
// stores, V^T rows in natural order) for the o_proj GEMM. Eight wavefronts = four windows per workgroup there.
// ------------------------------------------------------------------------------------------
template <int CG, int WS, int NW, bool FUSE_PROJ>
__global__ __launch_bounds__(NW * 64, 2) void swin_attn_block_x3_kernel(float *__restrict__ x, const float *__restrict__ gam,
                                                                    const float *__restrict__ bet, const char *__restrict__ wqkv,
                                                                    const float *__restrict__ bqkv, const char *__restrict__ wo,
                                                                    const float *__restrict__ bo,
                                                                    const float *__restrict__ bias_perm, char *__restrict__ ctx,
                                                                    WinGeom g, int total, float scale2, float eps) {
    constexpr int C = CG * 32, HEADS = CG, NWIN = NW / 2, CH = CG * 4096, NSTAGE = 3, PIECES = 4 * CG, PPW = PIECES / NW;
    constexpr int NS = 2 * CG, KPH = FUSE_PROJ ? 4 : 3, NCH = KPH * HEADS, KV = 64 * 128 + 2 * 32 * 128 + 64;
    static_assert(PIECES % NW == 0, "the wavefronts share the pieces of a chunk evenly");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int win = wave >> 1, qt = wave & 1;
    char *Ks = smem + NSTAGE * CH + win * KV;  // K: 64 keys x 128 B (chunks 0..3 hi, 4..7 lo, lds_off swizzle)
    char *Vh = Ks + 64 * 128, *Vl = Vh + 32 * 128;  // V hi / lo: [64 keys][32 dims] row-major, 64-byte rows
    unsigned char *Rg = (unsigned char *)(Vl + 32 * 128);
    float *bs = (float *)(smem + NSTAGE * CH + NWIN * KV);  // bq | bk | bv | bo

    int wid = blockIdx.x * NWIN + win;
    const bool wlive = wid < total;  // an odd window count leaves the last workgroup one idle pair (it keeps the barriers)
    wid = min(wid, total - 1);
    const int wlin = wid % g.nW, b = wid / g.nW;
    const int wy = wlin / g.nWx, wx = wlin - wy * g.nWx;
    const int ws = WS ? WS : g.ws;
    const int A = ws * ws;
    const int p = qt * 32 + r;  // window position = key index of this lane's token
    const bool valid = p < A;
    const size_t tok = win_token(g, ws, b, wy, wx, min(p, A - 1));

    for (int i = tid; i < 3 * C; i += NW * 64) bs[i] = bqkv[i];
    if (FUSE_PROJ)
        for (int i = tid; i < C; i += NW * 64) bs[3 * C + i] = bo[i];
    const bool masked = g.shift > 0 && (wy == g.H / ws - 1 || wx == g.nWx - 1);  // wave-uniform
    if (masked && qt == 0) Rg[lane] = (unsigned char)(lane < A ? win_region(g, ws, wy, wx, lane) : 0);

    int voffA[PPW], voffB[PPW];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj) {
            const int pc = jj * NW + wave;
            const int rho = (pc & 3) * 8 + lrow;  // q / k / v chunk: image pc >> 2 = k group, rows (pc & 3) * 8 .. + 7
            voffA[jj] = rho * (C * 4) + (pc >> 2) * 128 + ((slot ^ ((rho >> 1) & 7)) << 4);
            const int c = pc * 8 + lrow;  // o_proj chunk: rows pc * 8 .. + 7 of the C outputs, the head's k group
            voffB[jj] = c * (C * 4) + ((slot ^ ((c >> 1) & 7)) << 4);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const auto rsq = __builtin_amdgcn_make_buffer_rsrc((void *)wqkv, 0, (unsigned)(3 * C * C * 4), 0x00020000);
    const auto rso = __builtin_amdgcn_make_buffer_rsrc((void *)wo, 0, (unsigned)(C * C * 4), 0x00020000);
    // chunk j = KPH * head + kind (kind 0 .. 2: rows kind * C + 32 head .. + 31 of Wqkv; kind 3: column group `head` of Wo)
#define OCM_AB_DMA(j, st)                                                                                               \
    do {                                                                                                                \
        const int kind_ = (j) % KPH, head_ = (j) / KPH;                                                                 \
        if (kind_ < 3) {                                                                                                \
            _Pragma("unroll") for (int jj = 0; jj < PPW; ++jj)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (lds_ptr)(smem + (st) * CH + (jj * NW + wave) * 1024), 16, \
                                                         voffA[jj], (kind_ * C + head_ * 32) * (C * 4), 0, 0);          \
        } else {                                                                                                        \
            _Pragma("unroll") for (int jj = 0; jj < PPW; ++jj)                                                          \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rso, (lds_ptr)(smem + (st) * CH + (jj * NW + wave) * 1024), 16, \
                                                         voffB[jj], head_ * 128, 0, 0);                                 \
        }                                                                                                               \
    } while (0)
#else
#define OCM_AB_DMA(j, st) (void)0
#endif
    OCM_AB_DMA(0, 0);
    OCM_AB_DMA(1, 1);

    // LayerNorm of the wave's rows: lane (r, h) holds channels 16 s + 8 h .. + 7 of its token for every slice s
    bf16x8 xh[NS], xl[NS];
    {
        const float *xr = x + tok * C + 8 * h;
        f32x4 v[NS][2];
        float sum = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            v[sI][0] = *(const f32x4 *)(xr + 16 * sI);
            v[sI][1] = *(const f32x4 *)(xr + 16 * sI + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += v[sI][0][e] + v[sI][1][e];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float var = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = v[sI][0][e] - mean, d1 = v[sI][1][e] - mean;
                var = fmaf(d0, d0, fmaf(d1, d1, var));
            }
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * (1.0f / C) + eps);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const f32x4 g0 = *(const f32x4 *)(gam + 16 * sI + 8 * h), g1 = *(const f32x4 *)(gam + 16 * sI + 8 * h + 4);
            const f32x4 e0 = *(const f32x4 *)(bet + 16 * sI + 8 * h), e1 = *(const f32x4 *)(bet + 16 * sI + 8 * h + 4);
            split8((v[sI][0] - mean) * rstd * g0 + e0, (v[sI][1] - mean) * rstd * g1 + e1, xh[sI], xl[sI]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // rows, parameters and the first two chunks have landed (see swin_mlp_x3_kernel)

    f32x16 Y[FUSE_PROJ ? CG : 1];
#pragma unroll
    for (int mf = 0; mf < (FUSE_PROJ ? CG : 1); ++mf)
#pragma unroll
        for (int e = 0; e < 16; ++e) Y[mf][e] = 0.f;
    const int pr = pi_row(r);
    int sc = 0, si = NSTAGE - 1;
    // top of step j: own pieces of chunk j have landed (chunk j + 1 may still be on its way), everybody's have after the
    // barrier, and everybody is past the stage that chunk j + 2 goes into
    // (EXTRA: vector-memory loads issued between chunk j's DMA and chunk j + 1's, or after both — the head's eight bias loads)
#define OCM_AB_TOP(j, EXTRA)                                                           \
    do {                                                                               \
        if ((j) + 1 < NCH)                                                             \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + (EXTRA)) : "memory");       \
        else                                                                           \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        __builtin_amdgcn_s_barrier();                                                  \
        asm volatile("" ::: "memory");                                                 \
    } while (0)
#define OCM_AB_NEXT()                              \
    do {                                           \
        sc = sc == NSTAGE - 1 ? 0 : sc + 1;        \
        si = si == NSTAGE - 1 ? 0 : si + 1;        \
    } while (0)
    // 32 output features of the chunk in stage sc for the wave's 32 tokens: register e of lane half h = feature key_of_reg(e, h)
    auto project = [&](f32x16 &S) {
        const char *Wst = smem + sc * CH;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const char *img = Wst + (sI >> 1) * 4096;
            const bf16x8 ah = *(const bf16x8 *)(img + lds_off(pr, (sI & 1) * 2 + h));
            const bf16x8 al = *(const bf16x8 *)(img + lds_off(pr, 4 + (sI & 1) * 2 + h));
            S = mfma32x3(ah, al, xh[sI], xl[sI], S);
        }
    };
    // registers 8 s2 .. 8 s2 + 7 (features 16 s2 + 8 h .. + 7) + bias, as a pair of fragments
    auto biased_pair = [&](const f32x16 &S, const float *bvec, int s2, bf16x8 &ph, bf16x8 &pl) {
        const float *bp = bvec + 16 * s2 + 8 * h;
        const f32x4 c0 = *(const f32x4 *)bp, c1 = *(const f32x4 *)(bp + 4);
        f32x4 u0, u1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u0[e] = S[8 * s2 + e] + c0[e];
            u1[e] = S[8 * s2 + 4 + e] + c1[e];
        }
        split8(u0, u1, ph, pl);
    };
    const bf16x8 zero8 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};

    for (int head = 0; head < HEADS; ++head) {
        const int j0 = KPH * head;
        f32x16 S;
        bf16x8 qh[2], ql[2];
        // the head's relative-position bias (lane / register order of the score tiles), requested a whole head ahead of its use
        f32x4 bv[2][4];
        {
            const float *bq = bias_perm + ((size_t)head * 2 + qt) * 64 * 32 + lane * 4;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) bv[sub][e4] = *(const f32x4 *)(bq + (sub * 4 + e4) * 256);
        }
        // ---- q
        OCM_AB_TOP(j0, 8);
        OCM_AB_DMA(j0 + 2, si);
        project(S);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) biased_pair(S, bs + head * 32, s2, qh[s2], ql[s2]);
        OCM_AB_NEXT();
        // ---- k -> the window's K image, row p
        OCM_AB_TOP(j0 + 1, 8);
        if (j0 + 3 < NCH) OCM_AB_DMA(j0 + 3, si);
        project(S);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl;
            biased_pair(S, bs + C + head * 32, s2, ph, pl);
            *(bf16x8 *)(Ks + lds_off(p, 2 * s2 + h)) = valid ? ph : zero8;
            *(bf16x8 *)(Ks + lds_off(p, 4 + 2 * s2 + h)) = valid ? pl : zero8;
        }
        OCM_AB_NEXT();
        // ---- v -> the window's V images, row p
        OCM_AB_TOP(j0 + 2, 0);
        if (j0 + 4 < NCH) OCM_AB_DMA(j0 + 4, si);  // FUSE_PROJ: the next head's q chunk; otherwise its k chunk
        project(S);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl;
            biased_pair(S, bs + 2 * C + head * 32, s2, ph, pl);
            *(bf16x8 *)(Vh + p * 64 + (2 * s2 + h) * 16) = valid ? ph : zero8;
            *(bf16x8 *)(Vl + p * 64 + (2 * s2 + h) * 16) = valid ? pl : zero8;
        }
        OCM_AB_NEXT();
        // ---- scores, softmax, context (swin_wattn_x3_kernel for this wave's query tile), then y^T += Wo[:, head] . ctx^T
        if constexpr (FUSE_PROJ) {
            OCM_AB_TOP(j0 + 3, 0);
        } else {  // no chunk of its own: only the K / V images have to be complete
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        f32x16 S2[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S2[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 kh = *(const bf16x8 *)(Ks + lds_off(sub * 32 + pr, 2 * s + h));
                const bf16x8 kl = *(const bf16x8 *)(Ks + lds_off(sub * 32 + pr, 4 + 2 * s + h));
                S2[sub] = mfma32x3(kh, kl, qh[s], ql[s], S2[sub]);
            }
        }
        float mx = -INFINITY;
        const int myreg = masked ? Rg[min(p, A - 1)] : 0;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(S2[sub][e4 * 4 + e], scale2, bv[sub][e4][e]);
                    if (masked) {
                        const int jk = sub * 32 + key_of_reg(e4 * 4 + e, h);
                        if (jk < A && Rg[jk] != myreg) v += -100.0f * 1.4426950408889634f;
                    }
                    S2[sub][e4 * 4 + e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = fast_exp2(S2[sub][e] - mx);
                S2[sub][e] = pv;
                l += pv;
            }
        l += __shfl_xor(l, 32, 64);
        // the bias loads above are consumed: the next chunk's DMA is issued only now (hipcc drains every vector-memory
        // operation in front of the first use of an ordinary load's result while an LDS-DMA is in flight)
        if constexpr (FUSE_PROJ)
            if (j0 + 5 < NCH) OCM_AB_DMA(j0 + 5, si);
        f32x16 O;
#pragma unroll
        for (int e = 0; e < 16; ++e) O[e] = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = S2[sub][8 * s2 + e];
                    const bf16 t = (bf16)pv;
                    ph[e] = t;
                    pl[e] = (bf16)(pv - (float)t);
                }
                // V^T rows in pi order: lane 4 q + pp of a 16-lane group addresses key row q, dims 4 swap(pp) .. + 3, so that
                // column slot i of the group receives dim pi_row(i) and accumulator register e of lane half h holds dim
                // key_of_reg(e, h): registers 8 s .. 8 s + 7 are the B fragment (k = 16 s + 8 h ..) of the o_proj product
                const int pp = lane & 3, pps = FUSE_PROJ ? ((pp & 1) << 1) | (pp >> 1) : pp;
                const int voff = (sub * 32 + 16 * s2 + 8 * h + ((lane >> 2) & 3)) * 64 + (16 * ((lane >> 4) & 1) + 4 * pps) * 2;
                const bf16x8 vh = tr_read8(Vh + voff), vl = tr_read8(Vl + voff);
                O = mfma32x3(vh, vl, ph, pl, O);
            }
        if constexpr (!FUSE_PROJ) {
            // lane (r, h) holds dims {8 gq + 4 h + e} in fp32: the pair halves go out as four 8-byte pieces per half
            if (valid && wlive) {
                const float inv = 1.0f / l;
                char *dst = ctx + tok * ((size_t)C * 4) + head * 128;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = O[4 * gq + e] * inv;
                    bf16x4 oh, ol;
                    split4(o, oh, ol);
                    char *pd = dst + (8 * gq + 4 * h) * 2;
                    *(bf16x4 *)pd = oh;
                    *(bf16x4 *)(pd + 64) = ol;
                }
            }
        } else {
            const float inv = 1.0f / l;
            const char *Wos = smem + sc * CH;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = O[8 * s2 + e] * inv;
                    o1[e] = O[8 * s2 + 4 + e] * inv;
                }
                bf16x8 ch, cl;
                split8(o0, o1, ch, cl);
#pragma unroll
                for (int mf = 0; mf < CG; ++mf) {
                    const bf16x8 wh = *(const bf16x8 *)(Wos + lds_off(32 * mf + r, 2 * s2 + h));
                    const bf16x8 wl = *(const bf16x8 *)(Wos + lds_off(32 * mf + r, 4 + 2 * s2 + h));
                    Y[mf] = mfma32x3(wh, wl, ch, cl, Y[mf]);
                }
            }
        }
        if constexpr (FUSE_PROJ) OCM_AB_NEXT();
    }
#undef OCM_AB_DMA
#undef OCM_AB_TOP
#undef OCM_AB_NEXT
    if (!FUSE_PROJ || !valid || !wlive) return;
    // y^T: lane (r, h) register 4 gq + e of fragment mf = channel 32 mf + 8 gq + 4 h + e of the lane's token
    float *xo = x + tok * C;
#pragma unroll
    for (int mf = 0; mf < CG; ++mf)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = 32 * mf + 8 * gq + 4 * h;
            const f32x4 old = *(const f32x4 *)(xo + c), bb = *(const f32x4 *)(bs + 3 * C + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Y[mf][4 * gq + e] + bb[e] + old[e];
            *(f32x4 *)(xo + c) = o;
        }
}

// proj_fused: the whole half in one kernel (C = 96); otherwise (C = 192) the kernel stops at the context pairs
bool swin_attn_block_fused_supported(int prec, int C, int heads, int ws) {
    return prec == 2 && (C == 96 || C == 128 || C == 192) && heads * 32 == C && ws >= 2 && ws * ws <= 64;
}
bool swin_attn_block_proj_fused(int C) { return C == 96; }

template <int CG, int WS, int NW, bool FUSE_PROJ>
static hipError_t launch_swin_attn_block_t(float *x, const float *g, const float *be, const void *wqkv, const float *bqkv,
                                           const void *wo, const float *bo, const float *bias_perm, void *ctx, const WinGeom &gm,
                                           long total, float eps, hipStream_t s) {
    constexpr int C = CG * 32, NWIN = NW / 2;
    constexpr int lds = 3 * CG * 4096 + NWIN * (64 * 128 + 2 * 32 * 128 + 64) + 4 * C * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    const float scale2 = 0.17677669529663687f * 1.4426950408889634f;  // 32^-0.5 (SwinAttention.scaling :408) in the log2 domain
    const dim3 grid((unsigned)((total + NWIN - 1) / NWIN)), block(NW * 64);
    static unsigned long long optin = 0;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    auto kern = swin_attn_block_x3_kernel<CG, WS, NW, FUSE_PROJ>;
    if (!(optin >> (dev & 63) & 1)) {
        if (hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); e != hipSuccess)
            return e;
        optin |= 1ull << (dev & 63);
    }
    kern<<<grid, block, lds, s>>>(x, g, be, (const char *)wqkv, bqkv, (const char *)wo, bo, bias_perm, (char *)ctx, gm, (int)total,
                                  scale2, eps);
    return hipGetLastError();
}

// ctx: (tokens, C) split pairs, written when the projection is not fused (C = 192); unused otherwise
hipError_t launch_swin_attn_block(int prec, float *x, const float *g, const float *be, const void *wqkv, const float *bqkv,
                                  const void *wo, const float *bo, const float *bias_perm, void *ctx, int batch, int H, int W,
                                  int ws, int shift, int heads, int C, float eps, hipStream_t s) {
    if (!swin_attn_block_fused_supported(prec, C, heads, ws) || H % ws || W % ws || batch <= 0) return hipErrorInvalidValue;
    if (!swin_attn_block_proj_fused(C) && !ctx) return hipErrorInvalidValue;
    WinGeom gm{H, W, ws, shift, W / ws, (H / ws) * (W / ws), heads};
    const long total = (long)batch * gm.nW;
    if (total <= 0 || total > 0x3fffffffL) return hipErrorInvalidValue;
    if (C == 96)
        return ws == 7 ? launch_swin_attn_block_t<3, 7, 4, true>(x, g, be, wqkv, bqkv, wo, bo, bias_perm, ctx, gm, total, eps, s)
                       : launch_swin_attn_block_t<3, 0, 4, true>(x, g, be, wqkv, bqkv, wo, bo, bias_perm, ctx, gm, total, eps, s);
    if (C == 128)  // embed_dim 128 (the Swin-B family's first stage): up to the context, like C = 192
        return ws == 7 ? launch_swin_attn_block_t<4, 7, 8, false>(x, g, be, wqkv, bqkv, wo, bo, bias_perm, ctx, gm, total, eps, s)
                       : launch_swin_attn_block_t<4, 0, 8, false>(x, g, be, wqkv, bqkv, wo, bo, bias_perm, ctx, gm, total, eps, s);
    return ws == 7 ? launch_swin_attn_block_t<6, 7, 8, false>(x, g, be, wqkv, bqkv, wo, bo, bias_perm, ctx, gm, total, eps, s)
                   : launch_swin_attn_block_t<6, 0, 8, false>(x, g, be, wqkv, bqkv, wo, bo, bias_perm, ctx, gm, total, eps, s);
}

// ------------------------------------------------------------------------------------------
// final LayerNorm + AdaptiveAvgPool1d(1) + classifier: one workgroup per image
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swin_pool_head_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                             const float *__restrict__ be, const float *__restrict__ cw,
                                                             const float *__restrict__ cb, float *__restrict__ logits,
                                                             float *__restrict__ pooled, float *__restrict__ hidden,
                                                             int L, int C, int labels, float eps) {
    extern __shared__ float sm[];  // [4 waves][C] partial pooled sums, then [C] pooled
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *part = sm + wave * C;
    for (int c = lane; c < C; c += 64) part[c] = 0.f;
    for (int t = wave; t < L; t += 4) {
        const float *row = x + ((size_t)b * L + t) * C;
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += row[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)C;
        float var = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = row[c] - mean;
            var += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        const float rstd = rsqrtf(var / (float)C + eps);
        for (int c = lane; c < C; c += 64) {
            const float v = (row[c] - mean) * rstd * g[c] + be[c];
            if (hidden) hidden[((size_t)b * L + t) * C + c] = v;
            part[c] += v;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = (sm[c] + sm[C + c] + sm[2 * C + c] + sm[3 * C + c]) / (float)L;
        sm[4 * C + c] = v;
        if (pooled) pooled[(size_t)b * C + c] = v;
    }
    __syncthreads();
    const float *pv = sm + 4 * C;
    for (int o = wave; o < labels; o += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(pv[c], cw[(size_t)o * C + c], acc);
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) acc += __shfl_xor(acc, k, 64);
        if (lane == 0) logits[(size_t)b * labels + o] = acc + cb[o];
    }
}

// The same for C = 256 V channels (Swin-T: 768): eight wavefronts per image, a token's row as V 16-byte loads per lane (one
// round trip per token instead of three passes of 4-byte loads: 84 -> ~20 us at batch 256), the next token's row requested
// before the current one is reduced; statistics and the pooled sums in registers, partial sums combined in a fixed order.
template <int V>
__global__ __launch_bounds__(512) void swin_pool_head_vec_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                                 const float *__restrict__ be, const float *__restrict__ cw,
                                                                 const float *__restrict__ cb, float *__restrict__ logits,
                                                                 float *__restrict__ pooled, float *__restrict__ hidden,
                                                                 int L, int labels, float eps) {
    constexpr int C = 256 * V, NW = 8;
    extern __shared__ float sm[];  // [NW][C] partial pooled sums, then [C] pooled
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 gg[V], bb[V], acc[V], cur[V], nxt[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        gg[i] = *(const f32x4 *)(g + 4 * lane + 256 * i);
        bb[i] = *(const f32x4 *)(be + 4 * lane + 256 * i);
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        nxt[i] = acc[i];
    }
    if (wave < L) {
        const float *row = x + ((size_t)b * L + wave) * C + 4 * lane;
#pragma unroll
        for (int i = 0; i < V; ++i) nxt[i] = *(const f32x4 *)(row + 256 * i);
    }
    for (int t = wave; t < L; t += NW) {
#pragma unroll
        for (int i = 0; i < V; ++i) cur[i] = nxt[i];
        if (t + NW < L) {
            const float *row = x + ((size_t)b * L + t + NW) * C + 4 * lane;
#pragma unroll
            for (int i = 0; i < V; ++i) nxt[i] = *(const f32x4 *)(row + 256 * i);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) sum += (cur[i][0] + cur[i][1]) + (cur[i][2] + cur[i][3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)C;
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = cur[i][e] - mean;
                var = fmaf(d, d, var);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        const float rstd = rsqrtf(var / (float)C + eps);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (cur[i][e] - mean) * rstd * gg[i][e] + bb[i][e];
            if (hidden) *(f32x4 *)(hidden + ((size_t)b * L + t) * C + 4 * lane + 256 * i) = v;
            acc[i] += v;
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) *(f32x4 *)(sm + wave * C + 4 * lane + 256 * i) = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 512) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += sm[w * C + c];
        v /= (float)L;
        sm[NW * C + c] = v;
        if (pooled) pooled[(size_t)b * C + c] = v;
    }
    __syncthreads();
    const float *pv = sm + NW * C;
    for (int o = wave; o < labels; o += NW) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(pv[c], cw[(size_t)o * C + c], a);
#pragma unroll
        for (int k = 32; k > 0; k >>= 1) a += __shfl_xor(a, k, 64);
        if (lane == 0) logits[(size_t)b * labels + o] = a + cb[o];
    }
}

hipError_t launch_swin_pool_head(const float *x, const float *g, const float *be, const float *cw, const float *cb,
                                 float *logits, float *pooled, float *hidden, int batch, int L, int C, int labels,
                                 float eps, hipStream_t s) {
    const size_t ldsv = (size_t)9 * C * sizeof(float);
    if (C == 768)
        swin_pool_head_vec_kernel<3><<<dim3(batch), dim3(512), ldsv, s>>>(x, g, be, cw, cb, logits, pooled, hidden, L, labels, eps);
    else if (C == 1024)
        swin_pool_head_vec_kernel<4><<<dim3(batch), dim3(512), ldsv, s>>>(x, g, be, cw, cb, logits, pooled, hidden, L, labels, eps);
    else if (C == 512)
        swin_pool_head_vec_kernel<2><<<dim3(batch), dim3(512), ldsv, s>>>(x, g, be, cw, cb, logits, pooled, hidden, L, labels, eps);
    else if (C == 256)
        swin_pool_head_vec_kernel<1><<<dim3(batch), dim3(512), ldsv, s>>>(x, g, be, cw, cb, logits, pooled, hidden, L, labels, eps);
    else
        swin_pool_head_kernel<<<dim3(batch), dim3(256), 5 * C * sizeof(float), s>>>(x, g, be, cw, cb, logits, pooled, hidden,
                                                                                     L, C, labels, eps);
    return hipGetLastError();
}

// fp32 [rows][K] -> E [rows][Kp] with zero padding (weight upload; K padded to the GEMM's LDS row)
template <class E>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float *__restrict__ src, E *__restrict__ dst, size_t rows,
                                                       int K, int Kp) {
    const size_t total = rows * Kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / Kp;
        const int c = (int)(i - r * Kp);
        dst[i] = (E)(c < K ? src[r * K + c] : 0.f);
    }
}

hipError_t launch_cast_pad(int prec, const float *src, void *dst, size_t rows, int K, int Kp, hipStream_t s) {
    const size_t total = rows * Kp;
    unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (blocks == 0) return hipSuccess;
    if (prec == 2) {  // pair layout: rows are whole groups of 32, no padding to add
        if (K != Kp || K % 32) return hipErrorInvalidValue;
        return launch_cast_split(src, dst, total, s);
    }
    if (prec)
        cast_pad_kernel<float><<<dim3(blocks), dim3(256), 0, s>>>(src, (float *)dst, rows, K, Kp);
    else
        cast_pad_kernel<bf16><<<dim3(blocks), dim3(256), 0, s>>>(src, (bf16 *)dst, rows, K, Kp);
    return hipGetLastError();
}
