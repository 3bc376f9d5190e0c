// kernels_gemm.hip — the four GEMM-shaped stages of the ViT forward on MFMA:
//   patch embedding  (PatchEmbed.forward, dino/vision_transformer.py:129-132 + prepare_tokens :198-209)
//   qkv projection   (Attention.forward :80)
//   proj / fc2 + residual (Attention.forward :88, Block.forward :110-111, Mlp.forward :61)
//   fc1 + exact-erf GELU  (Mlp.forward :58-59)
// All share gemm_core.h's main loop; they differ in the A loader and the epilogue.
#include "gemm_core.h"
#include "launch.h"

// ------------------------------------------------------------------------------------------
// nn.Linear epilogues
// ------------------------------------------------------------------------------------------
template <int MODE>
struct EpiLinear {
    const float *bias;
    const float *resid;
    void *out;
    int M, N;
    int64_t ldo;
    __device__ __forceinline__ void operator()(const f32x16 &acc, int mb, int nb, int lane) const {
        if (nb >= N || mb >= M) return;
        const int n = nb + (lane & 31), h = lane >> 5;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = mb + acc_row32(reg, h);
            if (m < M) {
                const int64_t o = (int64_t)m * ldo + n;
                const float v = acc[reg] + bv;
                if (MODE == 0) ((float *)out)[o] = v;
                if (MODE == 1) ((float *)out)[o] = resid[o] + v;
                if (MODE == 2) ((bf16 *)out)[o] = (bf16)gelu_erf(v);
                if (MODE == 3) ((bf16 *)out)[o] = (bf16)v;
            }
        }
    }
};

template <class Cfg, bool SWAP, class ALoad, class Epi>
static hipError_t launch_gemm(const ALoad &al, const bf16 *w, int64_t ldw, int M, int N, int K, const Epi &epi,
                              hipStream_t s) {
    auto kern = gemm_kernel<Cfg, SWAP, ALoad, Epi>;
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN);
    kern<<<dim3(tiles), dim3(Cfg::NT), Cfg::LDS_BYTES, s>>>(al, w, ldw, M, N, K, epi);
    return hipGetLastError();
}

typedef GemmCfg<128, 128, 2, 2> Cfg128x128;
typedef GemmCfg<64, 128, 2, 2> Cfg64x128;
typedef GemmCfg<64, 64, 2, 2> Cfg64x64;

template <int MODE>
static hipError_t launch_linear_mode(const bf16 *a, const bf16 *w, const float *bias, const float *resid, void *out,
                                     int M, int N, int K, hipStream_t s) {
    RowLoader al{a, K};
    EpiLinear<MODE> epi{bias, resid, out, M, N, N};
    // Tile choice: fill >= 2 workgroups per CU (256 CUs) when the problem allows it.
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long t64 = (long)((M + 63) / 64) * ((N + 127) / 128);
    if (N % 128 == 0 && t128 >= 512) return launch_gemm<Cfg128x128, false>(al, w, K, M, N, K, epi, s);
    if (N % 128 == 0 && t64 >= 256) return launch_gemm<Cfg64x128, false>(al, w, K, M, N, K, epi, s);
    if (N % 128 == 0 && M > 64) return launch_gemm<Cfg64x128, false>(al, w, K, M, N, K, epi, s);
    return launch_gemm<Cfg64x64, false>(al, w, K, M, N, K, epi, s);
}

hipError_t launch_linear(const bf16 *a, const bf16 *w, const float *bias, const float *resid, void *out, int M,
                         int N, int K, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case 0: return launch_linear_mode<0>(a, w, bias, resid, out, M, N, K, s);
        case 1: return launch_linear_mode<1>(a, w, bias, resid, out, M, N, K, s);
        case 2: return launch_linear_mode<2>(a, w, bias, resid, out, M, N, K, s);
        case 3: return launch_linear_mode<3>(a, w, bias, resid, out, M, N, K, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// qkv projection -> head-major q, k and key-contiguous V^T
// ------------------------------------------------------------------------------------------
// Wqkv rows are ordered q(h0..hH-1), k(...), v(...), each head's 64 rows contiguous (:80).
// Column tiles inside [0, 2D) produce q/k rows  dst[(b*H + head)][t][d]          (d contiguous);
// column tiles inside [2D, 3D) run the main loop with the MFMA operands swapped, so the
// accumulator is transposed (lane = token) and V is written as V^T  vt[(b*H + head)][d][t]
// with the token index contiguous — the layout the P·V MFMA consumes — at full store width.
struct EpiQK {
    const float *bias;
    bf16 *q, *k;
    float *qkv32;  // optional (3,B,H,N,64) fp32, or nullptr
    int M, ntok, npad, H, D, B;
    __device__ __forceinline__ void operator()(const f32x16 &acc, int mb, int nb, int lane) const {
        if (mb >= M) return;
        const int which = nb / D, rem = nb - which * D;
        const int head = rem >> 6, d = (rem & 63) + (lane & 31), h = lane >> 5;
        const float bv = bias[nb + (lane & 31)];
        bf16 *dst = which ? k : q;
        const int b0 = mb / ntok, t0 = mb - b0 * ntok;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int dl = acc_row32(reg, h);
            if (mb + dl < M) {
                int t = t0 + dl, b = b0;
                while (t >= ntok) { t -= ntok; ++b; }
                const float v = acc[reg] + bv;
                dst[((int64_t)(b * H + head) * npad + t) * 64 + d] = (bf16)v;
                if (qkv32) qkv32[((((int64_t)which * B + b) * H + head) * ntok + t) * 64 + d] = v;
            }
        }
    }
};

struct EpiVt {
    const float *bias;
    bf16 *vt;
    float *qkv32;
    int M, ntok, npad, H, D, B;
    // acc is transposed: lane & 31 -> token row m, registers -> feature n.
    __device__ __forceinline__ void operator()(const f32x16 &acc, int mb, int nb, int lane) const {
        const int m = mb + (lane & 31), h = lane >> 5;
        if (m >= M) return;
        const int b = m / ntok, t = m - b * ntok;
        const int rem = nb - 2 * D, head = rem >> 6, dbase = rem & 63;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int dl = acc_row32(reg, h);
            const float v = acc[reg] + bias[nb + dl];
            const int d = dbase + dl;
            vt[((int64_t)(b * H + head) * 64 + d) * npad + t] = (bf16)v;
            if (qkv32) qkv32[((((int64_t)2 * B + b) * H + head) * ntok + t) * 64 + d] = v;
        }
    }
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void qkv_kernel(RowLoader al, const bf16 *__restrict__ W, int M, int D,
                                                      EpiQK eqk, EpiVt ev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = 3 * D, K = D;
    const int tiles_n = N / Cfg::BN;  // D % BN == 0 is checked by the launcher
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    f32x16 acc[Cfg::TM][Cfg::TN];
    if (n0 < 2 * D) {  // workgroup-uniform
        gemm_mainloop<Cfg, false>(al, W, K, m0, n0, M, N, K, smem, acc);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) eqk(acc[i][j], m0 + wm * Cfg::WM + 32 * i, n0 + wn * Cfg::WN + 32 * j, lane);
    } else {
        gemm_mainloop<Cfg, true>(al, W, K, m0, n0, M, N, K, smem, acc);
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) ev(acc[i][j], m0 + wm * Cfg::WM + 32 * i, n0 + wn * Cfg::WN + 32 * j, lane);
    }
}

template <class Cfg>
static hipError_t launch_qkv_cfg(const RowLoader &al, const bf16 *w, int M, int D, const EpiQK &eqk, const EpiVt &ev,
                                 hipStream_t s) {
    auto kern = qkv_kernel<Cfg>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * (3 * D / Cfg::BN);
    kern<<<dim3(tiles), dim3(Cfg::NT), Cfg::LDS_BYTES, s>>>(al, w, M, D, eqk, ev);
    return hipGetLastError();
}

hipError_t launch_qkv(const bf16 *a, const bf16 *w, const float *bias, bf16 *q, bf16 *k, bf16 *vt, float *qkv_f32,
                      int batch, int n_tokens, int n_pad, int heads, hipStream_t s) {
    const int D = heads * 64, M = batch * n_tokens;
    RowLoader al{a, D};
    EpiQK eqk{bias, q, k, qkv_f32, M, n_tokens, n_pad, heads, D, batch};
    EpiVt ev{bias, vt, qkv_f32, M, n_tokens, n_pad, heads, D, batch};
    const long t128 = (long)((M + 127) / 128) * (3 * D / 128);
    if (D % 128 == 0 && t128 >= 512) return launch_qkv_cfg<Cfg128x128>(al, w, M, D, eqk, ev, s);
    if (D % 128 == 0) return launch_qkv_cfg<Cfg64x128>(al, w, M, D, eqk, ev, s);
    return launch_qkv_cfg<Cfg64x64>(al, w, M, D, eqk, ev, s);
}

// ------------------------------------------------------------------------------------------
// patch embedding: im2col-free gather from fp32 planes + GEMM + bias + pos-embed
// ------------------------------------------------------------------------------------------
// Row m = b*P + py*wp + px is the p x p patch at (py, px) of tile b (row-major flatten, :131);
// column k = c*p*p + dy*p + dx indexes conv weight (D, C, p, p) flattened (:127). A 16-B LDS
// chunk is 8 consecutive dx of one (c, dy): two float4 loads from one image row, converted to
// bf16 on the way into LDS. Consecutive threads walk consecutive chunks of a row, so a wave
// reads whole 32-B..64-B row segments of neighbouring patches (coalesced along x).
struct PatchLoader {
    const float *image;
    int64_t sb, sc, sy;
    const int32_t *origins;
    int P, wp, p, pp;
    typedef const float *Handle;
    struct Raw {
        f32x4 lo, hi;
    };
    __device__ __forceinline__ Handle row(int m) const {
        const int b = m / P, pi = m - b * P;
        const int py = pi / wp, px = pi - py * wp;
        int y0 = 0, x0 = 0;
        if (origins) {
            y0 = origins[2 * b];
            x0 = origins[2 * b + 1];
        }
        return image + (int64_t)b * sb + (int64_t)(y0 + py * p) * sy + x0 + px * p;
    }
    __device__ __forceinline__ Raw load(Handle h, int k) const {
        const int c = k / pp, rem = k - c * pp;
        const int dy = rem / p, dx = rem - dy * p;
        const float *ptr = h + (int64_t)c * sc + (int64_t)dy * sy + dx;
        Raw r;
        r.lo = *(const f32x4 *)ptr;
        r.hi = *(const f32x4 *)(ptr + 4);
        return r;
    }
    __device__ __forceinline__ static bf16x8 finish(const Raw &r) { return cvt8(r.lo, r.hi); }
};

// x[b][1 + pi][n] = acc + bias[n] + pos[1 + pi][n]   (prepare_tokens :200-207, patch rows)
struct EpiPatch {
    const float *bias, *pos;
    float *x;
    int M, P, ntok, D;
    __device__ __forceinline__ void operator()(const f32x16 &acc, int mb, int nb, int lane) const {
        if (nb >= D || mb >= M) return;
        const int n = nb + (lane & 31), h = lane >> 5;
        const float bv = bias[n];
        const int b0 = mb / P, t0 = mb - b0 * P;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int dl = acc_row32(reg, h);
            if (mb + dl < M) {
                int t = t0 + dl, b = b0;
                while (t >= P) { t -= P; ++b; }
                x[((int64_t)b * ntok + 1 + t) * D + n] = acc[reg] + bv + pos[(int64_t)(1 + t) * D + n];
            }
        }
    }
};

hipError_t launch_patch_embed(const PatchArgs &pa, const bf16 *w, const float *bias, const float *pos, float *x,
                              int dim, hipStream_t s) {
    const int P = pa.hp * pa.wp, M = pa.batch * P, K = pa.chans * pa.p * pa.p;
    PatchLoader al{pa.image, pa.sb, pa.sc, pa.sy, pa.origins, P, pa.wp, pa.p, pa.p * pa.p};
    EpiPatch epi{bias, pos, x, M, P, P + 1, dim};
    if (dim % 128 == 0 && M > 64) return launch_gemm<Cfg64x128, false>(al, w, K, M, dim, K, epi, s);
    return launch_gemm<Cfg64x64, false>(al, w, K, M, dim, K, epi, s);
}
