// kernels_misc.hip — HBM-bound row kernels: LayerNorm, fp32->bf16 casts, cls rows.
#include "launch.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// nn.LayerNorm(D, eps) — reference: dino/vision_transformer.py:98,102,158 (eps 1e-6 via :262-278).
// One wavefront per row; the row lives in registers (float2 per lane per 128 columns), mean and
// biased variance are two in-register passes (same two-pass form as ATen's CPU kernel), fp32
// statistics. Output is bf16 (operand of the next MFMA GEMM) or fp32 (final norm -> feat).
// OUT: 0 = fp32, 1 = bf16, 2 = split-bf16 pairs (common.h: sp32 rows)
template <int OUT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, void *__restrict__ y,
                                                        int64_t rows, int dim, float eps) {
    constexpr int MAXV = 8;  // dim <= 1024
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * dim;
    f32x2 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 2 + i * 128;
        if (c < dim) {
            v[i] = *(const f32x2 *)(xr + c);
            s += v[i][0] + v[i][1];
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 2 + i * 128;
        if (c < dim) {
            const float a = v[i][0] - mean, b = v[i][1] - mean;
            q += a * a + b * b;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 2 + i * 128;
        if (c < dim) {
            const f32x2 g = *(const f32x2 *)(gamma + c), b = *(const f32x2 *)(beta + c);
            const float o0 = (v[i][0] - mean) * rstd * g[0] + b[0];
            const float o1 = (v[i][1] - mean) * rstd * g[1] + b[1];
            if (OUT == 1) {
                bf16x2 o;
                o[0] = (bf16)o0;
                o[1] = (bf16)o1;
                *(bf16x2 *)((bf16 *)y + row * dim + c) = o;
            } else if (OUT == 2) {
                bf16 h0, l0, h1, l1;
                split1(o0, h0, l0);
                split1(o1, h1, l1);
                bf16x2 hi, lo;
                hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
                char *rp = (char *)y + row * dim * 4 + sp_off(c);
                *(bf16x2 *)rp = hi;
                *(bf16x2 *)(rp + 64) = lo;
            } else {
                f32x2 o = {o0, o1};
                *(f32x2 *)((float *)y + row * dim + c) = o;
            }
        }
    }
}

// Fast path for dim % 128 == 0 (ViT-S 384, ViT-B 768): half a wavefront per row, float4 loads
// (dim/128 per lane, all issued before the first use), 8-byte bf16 / 16-byte fp32 stores.
template <int OUT, int NV>
__global__ __launch_bounds__(256) void layernorm_v4_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, void *__restrict__ y,
                                                           int64_t rows, float eps) {
    constexpr int dim = NV * 128;
    const int sub = threadIdx.x & 31;
    const int64_t row = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const float *xr = x + row * dim;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *(const f32x4 *)(xr + sub * 4 + i * 128);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * (1.0f / dim);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            q = fmaf(d, d, q);
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / dim) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = sub * 4 + i * 128;
        const f32x4 g = *(const f32x4 *)(gamma + c), b = *(const f32x4 *)(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        if (OUT == 1) {
            bf16x4 ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[e] = (bf16)o[e];
            *(bf16x4 *)((bf16 *)y + row * dim + c) = ob;
        } else if (OUT == 2) {
            bf16x4 hi, lo;
            split4(o, hi, lo);
            char *rp = (char *)y + row * dim * 4 + sp_off(c);
            *(bf16x4 *)rp = hi;
            *(bf16x4 *)(rp + 64) = lo;
        } else {
            *(f32x4 *)((float *)y + row * dim + c) = o;
        }
    }
}

template <int NV>
static hipError_t launch_ln_v4(const float *x, const float *gamma, const float *beta, void *y, int out_kind,
                               int64_t rows, float eps, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 7) / 8)), block(256);
    if (out_kind == 1)
        layernorm_v4_kernel<1, NV><<<grid, block, 0, s>>>(x, gamma, beta, y, rows, eps);
    else if (out_kind == 2)
        layernorm_v4_kernel<2, NV><<<grid, block, 0, s>>>(x, gamma, beta, y, rows, eps);
    else
        layernorm_v4_kernel<0, NV><<<grid, block, 0, s>>>(x, gamma, beta, y, rows, eps);
    return hipGetLastError();
}

// out_kind: OCM_LN_F32 (0) / OCM_LN_BF16 (1) / OCM_LN_SPLIT (2, dim % 32 == 0)
hipError_t launch_layernorm(const float *x, const float *gamma, const float *beta, void *y, int out_kind,
                            int64_t rows, int dim, float eps, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (out_kind == 2 && dim % 32) return hipErrorInvalidValue;
    switch (dim) {
        case 128: return launch_ln_v4<1>(x, gamma, beta, y, out_kind, rows, eps, s);
        case 256: return launch_ln_v4<2>(x, gamma, beta, y, out_kind, rows, eps, s);
        case 384: return launch_ln_v4<3>(x, gamma, beta, y, out_kind, rows, eps, s);
        case 512: return launch_ln_v4<4>(x, gamma, beta, y, out_kind, rows, eps, s);
        case 768: return launch_ln_v4<6>(x, gamma, beta, y, out_kind, rows, eps, s);
        case 1024: return launch_ln_v4<8>(x, gamma, beta, y, out_kind, rows, eps, s);
        default: break;
    }
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (out_kind == 1)
        layernorm_kernel<1><<<grid, block, 0, s>>>(x, gamma, beta, y, rows, dim, eps);
    else if (out_kind == 2)
        layernorm_kernel<2><<<grid, block, 0, s>>>(x, gamma, beta, y, rows, dim, eps);
    else
        layernorm_kernel<0><<<grid, block, 0, s>>>(x, gamma, beta, y, rows, dim, eps);
    return hipGetLastError();
}

// fp32 -> bf16 RNE (v_cvt_pk_bf16_f32), 8 elements per thread, grid-stride.
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float *__restrict__ src, bf16 *__restrict__ dst,
                                                        size_t count) {
    const size_t nvec = count >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4 *)(src + i * 8), b = *(const f32x4 *)(src + i * 8 + 4);
        *(bf16x8 *)(dst + i * 8) = cvt8(a, b);
    }
    // tail (count % 8) by the first threads of block 0
    if (blockIdx.x == 0) {
        const size_t t = (nvec << 3) + threadIdx.x;
        if (t < count) dst[t] = (bf16)src[t];
    }
}

hipError_t launch_cast_bf16(const float *src, bf16 *dst, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    size_t blocks = ((count >> 3) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    cast_bf16_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, dst, count);
    return hipGetLastError();
}

// fp32 -> split-bf16 pairs (common.h: sp32): every 32 consecutive elements become [32 x hi | 32 x lo] (128 bytes).
// `count` must be a multiple of 32 (all contraction extents here are). 8 elements per thread.
__global__ __launch_bounds__(256) void cast_split_kernel(const float *__restrict__ src, char *__restrict__ dst,
                                                         size_t count) {
    const size_t nvec = count >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4 *)(src + i * 8), b = *(const f32x4 *)(src + i * 8 + 4);
        bf16x8 hi, lo;
        split8(a, b, hi, lo);
        char *g = dst + (i >> 2) * 128 + (i & 3) * 16;
        *(bf16x8 *)g = hi;
        *(bf16x8 *)(g + 64) = lo;
    }
}

hipError_t launch_cast_split(const float *src, void *dst, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    if (count % 32) return hipErrorInvalidValue;
    size_t blocks = ((count >> 3) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    cast_split_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, (char *)dst, count);
    return hipGetLastError();
}

// inverse (tests / diagnostics): split pairs -> fp32 hi + lo
__global__ __launch_bounds__(256) void merge_split_kernel(const char *__restrict__ src, float *__restrict__ dst,
                                                          size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const char *g = src + (i >> 5) * 128 + (i & 31) * 2;
        dst[i] = (float)*(const bf16 *)g + (float)*(const bf16 *)(g + 64);
    }
}

hipError_t launch_merge_split(const void *src, float *dst, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    if (count % 32) return hipErrorInvalidValue;
    size_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    merge_split_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const char *)src, dst, count);
    return hipGetLastError();
}

// Grayscale fold of the patch-embedding conv weight: for R==G==B inputs
// conv(x, W) == conv(x[:, :1], W.sum(dim=1)). src (D, C, pp) fp32 -> dst (D, pp) bf16.
__global__ __launch_bounds__(256) void fold_cast_kernel(const float *__restrict__ src, bf16 *__restrict__ dst, int D,
                                                        int C, int pp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * pp) return;
    const int d = i / pp, k = i - d * pp;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += src[((size_t)d * C + c) * pp + k];
    dst[i] = (bf16)s;
}

__global__ __launch_bounds__(256) void fold_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, int D,
                                                       int C, int pp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * pp) return;
    const int d = i / pp, k = i - d * pp;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += src[((size_t)d * C + c) * pp + k];
    dst[i] = s;
}

__global__ __launch_bounds__(256) void fold_split_kernel(const float *__restrict__ src, char *__restrict__ dst, int D,
                                                         int C, int pp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * pp) return;
    const int d = i / pp, k = i - d * pp;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += src[((size_t)d * C + c) * pp + k];
    bf16 hi, lo;
    split1(s, hi, lo);
    char *g = dst + (size_t)(i >> 5) * 128 + (i & 31) * 2;  // pp % 32 == 0: groups never straddle rows
    *(bf16 *)g = hi;
    *(bf16 *)(g + 64) = lo;
}

hipError_t launch_fold_split(const float *src, void *dst, int D, int C, int pp, hipStream_t s) {
    const int n = D * pp;
    if (pp % 32) return hipErrorInvalidValue;
    fold_split_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(src, (char *)dst, D, C, pp);
    return hipGetLastError();
}

hipError_t launch_fold_f32(const float *src, float *dst, int D, int C, int pp, hipStream_t s) {
    const int n = D * pp;
    fold_f32_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(src, dst, D, C, pp);
    return hipGetLastError();
}

hipError_t launch_fold_cast_bf16(const float *src, bf16 *dst, int D, int C, int pp, hipStream_t s) {
    const int n = D * pp;
    fold_cast_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(src, dst, D, C, pp);
    return hipGetLastError();
}

// prepare_tokens (:203-207), cls part: x[b][0][:] = cls_token + pos_embed[0].
__global__ __launch_bounds__(256) void cls_rows_kernel(const float *__restrict__ cls, const float *__restrict__ pos,
                                                       float *__restrict__ x, int batch, int n_tokens, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * dim) return;
    const int b = i / dim, c = i - b * dim;
    x[(size_t)b * n_tokens * dim + c] = cls[c] + pos[c];
}

hipError_t launch_cls_rows(const float *cls, const float *pos, float *x, int batch, int n_tokens, int dim,
                           hipStream_t s) {
    const int n = batch * dim;
    cls_rows_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(cls, pos, x, batch, n_tokens, dim);
    return hipGetLastError();
}

// ---- LayerNorm folded into the GEMM that consumes it (split-bf16 forward) ----
// The residual stream x is handed to the next GEMM UN-normalised, as split pairs, next to per-row sums (S1 = sum x,
// S2 = sum x^2, accumulated by the producing epilogues); the consumer multiplies by W' = W * gamma and finishes
//   LN(x) W^T + b = rstd * (x W'^T - mu * c) + d,   c[n] = sum_k W'[n][k],   d[n] = sum_k beta[k] W[n][k] + b[n]
// in its epilogue. The cls row of every image (prepare_tokens :203-207) is produced here with its sums (one wavefront per
// image; the patch rows come from the patch-embedding epilogue).
__global__ __launch_bounds__(64) void cls_rows_stats_kernel(const float *__restrict__ cls, const float *__restrict__ pos,
                                                            float *__restrict__ x, char *__restrict__ xs,
                                                            float *__restrict__ stats, int n_tokens, int dim, int batch,
                                                            float *__restrict__ shift, float *__restrict__ tok_shift,
                                                            const float *__restrict__ pe_bias) {
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= batch) {  // blocks batch .. batch + n_tokens - 1: the patch rows' centring constants (launch.h)
        const int n = (int)blockIdx.x - batch;
        float sm = 0.f;
        for (int c = lane; c < dim; c += 64) sm += pos[(size_t)n * dim + c] + (pe_bias ? pe_bias[c] : 0.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        if (lane == 0) tok_shift[n] = sm / (float)dim;
        return;
    }
    const int b = blockIdx.x;
    const size_t row = (size_t)b * n_tokens;
    float sh = 0.f;
    if (shift) {  // the cls row's exact mean
        for (int c = lane; c < dim; c += 64) sh += cls[c] + pos[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sh += __shfl_xor(sh, o, 64);
        sh /= (float)dim;
        if (lane == 0) shift[row] = sh;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < dim; c += 64) {
        const float v = cls[c] + pos[c];
        x[row * dim + c] = v;
        const float vc = v - sh;
        bf16 hi, lo;
        split1(vc, hi, lo);
        char *g = xs + row * dim * 4 + sp_off(c);
        *(bf16 *)g = hi;
        *(bf16 *)(g + 64) = lo;
        s1 += vc;
        s2 = fmaf(vc, vc, s2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    const int nslot = dim >> 6;  // slot 0 carries the sums of the whole row, the others zeros (launch.h: LnFold)
    if (lane < nslot) {
        stats[(row * nslot + lane) * 2] = lane ? 0.f : s1;
        stats[(row * nslot + lane) * 2 + 1] = lane ? 0.f : s2;
    }
}

hipError_t launch_cls_rows_stats(const float *cls, const float *pos, float *x, void *xs, float *stats, int batch,
                                 int n_tokens, int dim, hipStream_t s, float *shift, float *tok_shift, const float *pe_bias) {
    const int extra = (shift && tok_shift) ? n_tokens : 0;
    cls_rows_stats_kernel<<<dim3(batch + extra), dim3(64), 0, s>>>(cls, pos, x, (char *)xs, stats, n_tokens, dim, batch,
                                                                   extra ? shift : nullptr, tok_shift, pe_bias);
    return hipGetLastError();
}

// Split-K finish (launch.h: StatsOut::part): one wavefront per output row adds the partial sums of the K slices in slice
// order to bias + residual, writes x (fp32, may alias resid) and, for a folded LayerNorm downstream, the split pairs and
// the row sums of x in the slot layout of EpiResidStats (slot 0 = sums of the whole row, the other slots zero).
// SLICES > 0 and N <= 256 * NV: every load of a row (residual, bias, SLICES partial sums per chunk, the previous site's slots) is
// issued before the first use — one L2 round trip instead of a chain of them (this kernel is ~200 rows of a one-tile forward:
// 5.9 -> 3.x us per launch, eleven launches per forward). SLICES = 0: any slice count / width, loads in a loop.
template <int SLICES, int NV>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float *__restrict__ part, int slices,
                                                            const float *__restrict__ bias, const float *resid,
                                                            float *x, char *__restrict__ xs, float *__restrict__ stats,
                                                            int M, int N, float *__restrict__ shift,
                                                            const float *__restrict__ prev_stats,
                                                            const float *__restrict__ prev_shift) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const size_t row = (size_t)m * N, slice = (size_t)M * N;
    const int nslot = N >> 6;
    float ps = 0.f, psh = 0.f;
    if (shift) {  // requested first: the row's mean at the previous LayerNorm site (launch.h: row centring)
        if (prev_stats)
            for (int i = lane; i < nslot; i += 64) ps += prev_stats[((size_t)m * nslot + i) * 2];
        if (prev_shift) psh = prev_shift[m];
    }
    f32x4 v[SLICES ? NV : 1];
    if constexpr (SLICES > 0) {
        f32x4 pt[NV][SLICES], bs[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = min(lane * 4 + i * 256, N - 4);
            v[i] = *(const f32x4 *)(resid + row + c);
            bs[i] = bias ? *(const f32x4 *)(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < SLICES; ++k) pt[i][k] = *(const f32x4 *)(part + k * slice + row + c);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] += bs[i];
#pragma unroll
            for (int k = 0; k < SLICES; ++k) v[i] += pt[i][k];  // slice order, as the loop form
        }
    }
    float sh = 0.f;
    if (shift) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
        sh = psh + ps / (float)N;
        if (lane == 0) shift[m] = sh;
    }
    float s1 = 0.f, s2 = 0.f;
    auto emit = [&](int c, f32x4 val) {
        *(f32x4 *)(x + row + c) = val;
        val -= sh;
        if (xs) {
            bf16x4 hi, lo;
            split4(val, hi, lo);
            char *g = xs + row * 4 + sp_off(c);
            *(bf16x4 *)g = hi;
            *(bf16x4 *)(g + 64) = lo;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1 += val[e];
            s2 = fmaf(val[e], val[e], s2);
        }
    };
    if constexpr (SLICES > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane * 4 + i * 256 < N) emit(lane * 4 + i * 256, v[i]);
    } else {
        for (int c = lane * 4; c < N; c += 256) {
            f32x4 val = *(const f32x4 *)(resid + row + c);
            if (bias) val += *(const f32x4 *)(bias + c);
            for (int k = 0; k < slices; ++k) val += *(const f32x4 *)(part + k * slice + row + c);
            emit(c, val);
        }
    }
    if (!stats) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane < nslot) {
        stats[((size_t)m * nslot + lane) * 2] = lane ? 0.f : s1;
        stats[((size_t)m * nslot + lane) * 2 + 1] = lane ? 0.f : s2;
    }
}

hipError_t launch_splitk_finish(const float *part, int slices, const float *bias, const float *resid, float *x, void *xs,
                                float *stats, int M, int N, hipStream_t s, const StatsOut &so) {
    if (N % 4 || (stats && (N % 64 || N / 64 > 64)) || slices <= 0 || M <= 0) return hipErrorInvalidValue;
    const dim3 g((M + 3) / 4), b(256);
    float *sh = stats ? so.shift : nullptr;
    if (slices == OCM_SPLITK && N <= 512)  // ViT-T / ViT-S widths: the unrolled form
        splitk_finish_kernel<OCM_SPLITK, 2><<<g, b, 0, s>>>(part, slices, bias, resid, x, (char *)xs, stats, M, N, sh,
                                                            so.prev_stats, so.prev_shift);
    else
        splitk_finish_kernel<0, 1><<<g, b, 0, s>>>(part, slices, bias, resid, x, (char *)xs, stats, M, N, sh, so.prev_stats,
                                                   so.prev_shift);
    return hipGetLastError();
}

// W (N, K) fp32, gamma / beta (K), bias (N) -> W' = W * gamma as split pairs, c (N), d (N). One wavefront per output row;
// the two row sums in float64 (they multiply O(1) row statistics in every epilogue).
__global__ __launch_bounds__(64) void fold_ln_kernel(const float *__restrict__ W, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, const float *__restrict__ bias,
                                                     char *__restrict__ Wf, float *__restrict__ cvec,
                                                     float *__restrict__ dvec, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float *w = W + (size_t)n * K;
    char *dst = Wf + (size_t)n * K * 4;
    double c = 0.0, d = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float wg = w[k] * gamma[k];
        bf16 hi, lo;
        split1(wg, hi, lo);
        *(bf16 *)(dst + sp_off(k)) = hi;
        *(bf16 *)(dst + sp_off(k) + 64) = lo;
        c += (double)wg;
        d += (double)beta[k] * (double)w[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o, 64);
        d += __shfl_xor(d, o, 64);
    }
    if (lane == 0) {
        cvec[n] = (float)c;
        dvec[n] = (float)(d + (double)(bias ? bias[n] : 0.f));
    }
}

hipError_t launch_fold_ln(const float *W, const float *gamma, const float *beta, const float *bias, void *Wf, float *cvec,
                          float *dvec, int N, int K, hipStream_t s) {
    if (K % 32) return hipErrorInvalidValue;
    fold_ln_kernel<<<dim3(N), dim3(64), 0, s>>>(W, gamma, beta, bias, (char *)Wf, cvec, dvec, K);
    return hipGetLastError();
}

// ---- encoders of model.py:48-53,134-139: x[:, 1:].permute(0, 2, 1).reshape(B, C, H, W) ----
// y: (B, N, D) normed tokens; out: (B, D, N-1). 32x32 tiles through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void tokens_to_fmap_kernel(const float *__restrict__ y, float *__restrict__ out,
                                                             int N, int D) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int P = N - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i;
        if (t < P) tile[ty + 8 * i][tx] = y[((size_t)b * N + 1 + t) * D + d0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = d0 + ty + 8 * i, t = t0 + tx;
        if (t < P) out[((size_t)b * D + d) * P + t] = tile[tx][ty + 8 * i];
    }
}

hipError_t launch_tokens_to_fmap(const float *y, float *out, int batch, int n_tokens, int dim, hipStream_t s) {
    const dim3 grid((n_tokens - 1 + 31) / 32, dim / 32, batch);
    tokens_to_fmap_kernel<<<grid, dim3(256), 0, s>>>(y, out, n_tokens, dim);
    return hipGetLastError();
}

// ---- nn.PixelShuffle(s) of a token-major 1x1 conv (model.py:60-66): one thread per output pixel quad ----
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float *__restrict__ lin, float *__restrict__ out,
                                                            int hp, int wp, int c_out, int sh, size_t total) {
    const int Hs = hp * sh, Ws = wp * sh, O = c_out * sh * sh;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xo = (int)(i % Ws);
        size_t r = i / Ws;
        const int yo = (int)(r % Hs);
        r /= Hs;
        const int c = (int)(r % c_out), b = (int)(r / c_out);
        const int y = yo / sh, ii = yo - y * sh, x = xo / sh, jj = xo - x * sh;
        out[i] = lin[((size_t)b * hp * wp + (size_t)y * wp + x) * O + (c * sh + ii) * sh + jj];
    }
}

hipError_t launch_pixel_shuffle(const float *lin, float *out, int batch, int hp, int wp, int c_out, int sh, hipStream_t s) {
    const size_t total = (size_t)batch * c_out * hp * sh * wp * sh;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    pixel_shuffle_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(lin, out, hp, wp, c_out, sh, total);
    return hipGetLastError();
}

// ---- 3x3 / padding 1 convolution as a GEMM (LinearProbing's two-layer decoder, model.py:154-166) ----
// in: token-major fp32 [B][h*w][C]; out: operand rows [B*h*w][9*C] with K index (ky*3 + kx)*C + c, zero outside the
// image, optional ReLU on the way (the nn.ReLU between the two convolutions). One thread = 8 consecutive channels
// of one (token, tap); C % 32 == 0 keeps every 8-channel run inside one split-pair group.
template <int OUT>  // 0 fp32, 1 bf16, 2 split pairs
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float *__restrict__ in, char *__restrict__ out, int B, int h,
                                                        int w, int C, int relu) {
    const int c8 = C >> 3;
    const size_t total = (size_t)B * h * w * 9 * c8;
    const int esz = OUT == 1 ? 2 : 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        size_t r = i / c8;
        const int tap = (int)(r % 9);
        r /= 9;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h), b = (int)(r / h);
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const float *src = in + (((size_t)b * h + yy) * w + xx) * C + cc * 8;
            v0 = *(const f32x4 *)src;
            v1 = *(const f32x4 *)(src + 4);
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = fmaxf(v0[e], 0.f);
                    v1[e] = fmaxf(v1[e], 0.f);
                }
            }
        }
        const size_t row = ((size_t)b * h + y) * w + x;
        char *rowp = out + row * (size_t)(9 * C) * esz;
        const int col = tap * C + cc * 8;
        if (OUT == 1) {
            *(bf16x8 *)(rowp + col * 2) = cvt8(v0, v1);
        } else if (OUT == 0) {
            *(f32x4 *)(rowp + col * 4) = v0;
            *(f32x4 *)(rowp + col * 4 + 16) = v1;
        } else {
            bf16x8 hi, lo;
            split8(v0, v1, hi, lo);
            *(bf16x8 *)(rowp + sp_off(col)) = hi;
            *(bf16x8 *)(rowp + sp_off(col) + 64) = lo;
        }
    }
}

hipError_t launch_im2col3x3(int prec, const float *in, void *out, int batch, int h, int w, int C, int relu, hipStream_t s) {
    if (C % 32 || batch <= 0 || h <= 0 || w <= 0) return hipErrorInvalidValue;
    const size_t total = (size_t)batch * h * w * 9 * (C >> 3);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)blocks), block(256);
    if (prec == 0)
        im2col3x3_kernel<1><<<grid, block, 0, s>>>(in, (char *)out, batch, h, w, C, relu);
    else if (prec == 1)
        im2col3x3_kernel<0><<<grid, block, 0, s>>>(in, (char *)out, batch, h, w, C, relu);
    else
        im2col3x3_kernel<2><<<grid, block, 0, s>>>(in, (char *)out, batch, h, w, C, relu);
    return hipGetLastError();
}
