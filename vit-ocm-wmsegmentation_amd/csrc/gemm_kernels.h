// gemm_kernels.h — the GEMM-shaped stages of the ViT forward on MFMA: kernels, epilogues and per-element-type launchers
//   patch embedding  (PatchEmbed.forward, dino/vision_transformer.py:129-132 + prepare_tokens :198-209)
//   qkv projection   (Attention.forward :80)
//   proj / fc2 + residual [+ the LayerNorm that follows] (Attention.forward :88, Block.forward :110-111, Mlp.forward :61)
//   fc1 + exact-erf GELU  (Mlp.forward :58-59)
// All share gemm_core.h's main loops; they differ in the A loader and the epilogue. Everything here is a template on the
// operand element type E (bf16 / float / sp32 = split-bf16 pairs). The five entry templates launch_*_e<E> are
// explicitly instantiated, one element type and one half of the entry points per translation unit
// (kernels_gemm_inst.hip, compiled six times by the Makefile so that the library builds in parallel); the precision
// switches that call them live in kernels_gemm.hip.
#pragma once
#include <type_traits>
#include "gemm_core.h"
#include "launch.h"
#include "dev_knobs.h"

int ocm_wt_mask();  // kernels_gemm.hip

// LDS-DMA ring depth of the 64 x 128 tiles that serve the one-tile-per-call forwards (M <= 1024 rows: a handful of
// workgroups whose K loop is a chain of L2 round trips — three steps in flight instead of one)
#ifndef OCM_SMALLM_STAGES
#define OCM_SMALLM_STAGES 4
#endif
// ... and up to how many rows they are chosen: 4096 covers one ViT-S/8 window of 384^2 per call (2305 rows, 111 .. 444
// workgroups). With FOUR waves per tile the register-staged loop was faster there (attn.qkv 18.8 against 21.8 us, mlp.fc1 16.2
// against 17.5); with eight the LDS-DMA loop wins (the one-window forward 1.42 -> 1.28 ms)
constexpr int OCM_SMALLM_ROWS = 4096;
typedef GemmCfg<128, 128, 2, 2> Cfg128x128;
// the qkv projection runs the same tile with 8 waves (32x64 MFMA sub-tiles per wave): two waves per SIMD inside
// one workgroup overlap its heavier scatter epilogue with the other waves' MFMAs (29.0 -> 25.8 us at ViT-S, B=64)
typedef GemmCfg<128, 128, 2, 4> Cfg128x128q;
// ... and on v_mfma_f32_16x16x32_bf16 for the split-bf16 LDS-DMA launches of that tile (gemm_core.h: GemmCfg::MF16)
typedef GemmCfg<128, 128, 2, 4, 1> Cfg128x128q16;
typedef GemmCfg<128, 128, 2, 2, 1> Cfg128x128m16;  // development A/B (four waves: the tile of forwards of 32 k rows and more)
// 160 rows: two full 32-row tiles and a 16-row half tile per wave row (gemm_core.h: GemmCfg::HALF). At 12 608 rows (ViT-S/16, B = 64)
// mlp.fc1 is 79 x 12 = 948 of these = 1.85 rounds of the 512 two-per-CU slots instead of 1 188 tiles of 128 x 128 = 2.32
typedef GemmCfg<160, 128, 2, 4, 1> Cfg160x128q16;
typedef GemmCfg<256, 256, 2, 4, 1> Cfg256x256m16;  // ViT-B sizes
typedef GemmCfg<64, 128, 2, 2> Cfg64x128;
// the same tile on eight wavefronts (32 x 32 each) for the one-tile-per-call forwards: with ONE workgroup per CU a lone wave per
// SIMD waits out every LDS read before its MFMAs (850 cycles per K step for 384 of MFMA, tools/stamps_b1.py); two waves per SIMD
// take turns
typedef GemmCfg<64, 128, 2, 4> Cfg64x128w;
typedef GemmCfg<64, 64, 2, 2> Cfg64x64;
// 8 waves, one workgroup per CU: half the L2->LDS bytes per output element of 128x128. Pays off once the
// problem has at least two full rounds of such tiles (ViT-B at 384^2, the ViT-S/8 slab windows); below
// that the idle CUs of the last round cost more than the traffic saves, and with K = 384 (six steps) the
// exposed prologue of a lone workgroup does (measured: ViT-S/8 slab fc1 +7 % slower, ViT-B GEMMs 13 % faster).
typedef GemmCfg<256, 256, 2, 4> Cfg256x256;

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember per kernel template
// instantiation (one static mask each) on which devices it has been set.
static inline hipError_t ensure_lds_optin(const void *kern, int bytes, unsigned long long &done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask & bit) return hipSuccess;  // benign race: setting twice is idempotent
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done_mask |= bit;
    return e;
}

static inline bool big_tiles_pay(int M, int N, int K) {
    return K >= 768 && N % 256 == 0 && (long)((M + 255) / 256) * (N / 256) >= 512;
}

// Store Elem<OE>::EPW consecutive activations (taken from fp32 values) at element column `col` of the row that
// starts at `rowp`: 8 bf16 (16 B), 4 fp32 (16 B) or 8 split pairs (16 B of hi halves + 16 B of lo halves).
__device__ __forceinline__ void store_act(bf16 *, char *rowp, int col, const f32x4 &v0, const f32x4 &v1) {
    *(bf16x8 *)(rowp + col * 2) = cvt8(v0, v1);
}
__device__ __forceinline__ void store_act(float *, char *rowp, int col, const f32x4 &v0, const f32x4 &) {
    *(f32x4 *)(rowp + col * 4) = v0;
}
__device__ __forceinline__ void store_act(sp32 *, char *rowp, int col, const f32x4 &v0, const f32x4 &v1) {
    bf16x8 hi, lo;
    split8(v0, v1, hi, lo);
    *(bf16x8 *)(rowp + sp_off(col)) = hi;
    *(bf16x8 *)(rowp + sp_off(col) + 64) = lo;
}
// Write-through (sc1) 16-byte store: the line leaves the XCD's L2 instead of staying in it. For outputs that are far
// larger than the L2 and are not re-read by this kernel (fc1's 77 MB hidden tensor), plain stores evict the operands
// the other workgroups of the XCD are still streaming. The trailing s_nop covers the store-data hazard hipcc cannot
// see inside an asm statement.
__device__ __forceinline__ void store16_wt(void *p, const bf16x8 &v) {
    const f32x4 d = __builtin_bit_cast(f32x4, v);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ void store_act_wt(sp32 *, char *rowp, int col, const f32x4 &v0, const f32x4 &v1) {
    bf16x8 hi, lo;
    split8(v0, v1, hi, lo);
    store16_wt(rowp + sp_off(col), hi);
    store16_wt(rowp + sp_off(col) + 64, lo);
}
template <class OE>
__device__ __forceinline__ void store_act_wt(OE *t, char *rowp, int col, const f32x4 &v0, const f32x4 &v1) {
    store_act(t, rowp, col, v0, v1);
}
// one element
__device__ __forceinline__ void store_act1(bf16 *, char *rowp, int col, float v) { *(bf16 *)(rowp + col * 2) = (bf16)v; }
__device__ __forceinline__ void store_act1(float *, char *rowp, int col, float v) { *(float *)(rowp + col * 4) = v; }
__device__ __forceinline__ void store_act1(sp32 *, char *rowp, int col, float v) {
    bf16 hi, lo;
    split1(v, hi, lo);
    *(bf16 *)(rowp + sp_off(col)) = hi;
    *(bf16 *)(rowp + sp_off(col) + 64) = lo;
}

// ------------------------------------------------------------------------------------------
// nn.Linear epilogues
// ------------------------------------------------------------------------------------------
// out = epilogue(acc) with the bias already in the accumulator (gemm_mainloop).
//   MODE 0: fp32 out            MODE 1: fp32 out = resid + acc (resid may alias out)
//   MODE 2: OE out = gelu(acc)  MODE 3: OE out = acc          (OE = operand type of the next GEMM)
// Every lane moves 16 B per chunk. The residual of ALL of a lane's chunks is requested in one burst
// before the first use (one exposed L2 round trip instead of one per unrolled group).
// (mu, rstd) of row m from the sums its producers accumulated (launch.h: LnFold)
// Folded-LayerNorm helpers (launch.h: LnFold). The row sums of row m: all slot loads in flight at once (clamped indices, no
// run-time trip count: a loop would wait for every load in turn), added in index order LATER (ln_row_final): deterministic.
__device__ __forceinline__ void ln_row_raw(const LnFold &ln, int64_t m, f32x2 (&raw)[EPI_MAXS]) {
    if (!ln.stats) return;
    const f32x2 *sp = (const f32x2 *)ln.stats + m * ln.nslot;
    if (ln.nslot <= EPI_MAXS) {
        const int last = ln.nslot - 1;
#pragma unroll
        for (int i = 0; i < EPI_MAXS; ++i) {
            const f32x2 v = sp[min(i, last)];
            raw[i] = i <= last ? v : f32x2{0.f, 0.f};
        }
    } else {  // wide rows (ViT-B: 12 slots): summed here
        f32x2 sm = sp[0];
        for (int i = 1; i < ln.nslot; ++i) sm += sp[i];
        raw[0] = sm;
    }
}
__device__ __forceinline__ f32x2 ln_row_final(const LnFold &ln, const f32x2 (&raw)[EPI_MAXS]) {
    if (!ln.stats) return f32x2{0.f, 1.f};
    f32x2 sm = raw[0];
#pragma unroll
    for (int i = 1; i < EPI_MAXS; ++i) sm += raw[i];
    const float mu = sm[0] * ln.inv_dim;
    const float var = fmaxf(fmaf(-mu, mu, sm[1] * ln.inv_dim), 0.f);
    return f32x2{mu, 1.0f / sqrtf(var + ln.eps)};
}
__device__ __forceinline__ f32x2 ln_col_entry(const LnFold &ln, int n) {
    return ln.stats ? f32x2{ln.c[n], ln.d[n]} : f32x2{0.f, 0.f};
}

template <int MODE, class OE>
struct EpiLinear {
    const float *bias;
    const float *resid;
    void *out;
    int M, N;
    int64_t ldo;
    int wt = 0;  // activation outputs with write-through (sc1) stores
    LnFold ln;   // MODE 2 / 3: the LayerNorm in front of this layer, folded (the accumulator then starts at 0, not at the bias)
    static constexpr bool ROWTAB = (MODE == 2 || MODE == 3);
    __device__ __forceinline__ void row_raw(int m, f32x2 (&raw)[EPI_MAXS]) const { ln_row_raw(ln, m, raw); }
    __device__ __forceinline__ f32x2 row_final(const f32x2 (&raw)[EPI_MAXS]) const { return ln_row_final(ln, raw); }
    __device__ __forceinline__ f32x2 col_entry(int n) const { return ln_col_entry(ln, n); }
    template <class Cfg>
    __device__ __forceinline__ void run(const float *C, int m0, int n0, const f32x2 *rowtab, const f32x2 *coltab) const {
        constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
        constexpr bool ACT_OUT = (MODE == 2 || MODE == 3);
        constexpr int W = ACT_OUT ? Elem<OE>::EPW : 4;  // columns per lane
        constexpr int CPR = BN / W;                            // chunks per row
        constexpr int TOTAL = BM * CPR, ITERS = (TOTAL + NT - 1) / NT;
        f32x4 rs[MODE == 1 ? ITERS : 1];
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int q = min((int)threadIdx.x + i * NT, TOTAL - 1), row = q / CPR, col = (q - row * CPR) * W;
                rs[i] = *(const f32x4 *)(resid + (int64_t)min(m0 + row, M - 1) * ldo + min(n0 + col, N - W));
            }
        }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int q = threadIdx.x + i * NT, row = q / CPR, col = (q - row * CPR) * W;
            const int m = m0 + row, n = n0 + col;
            if ((TOTAL % NT != 0 && q >= TOTAL) || m >= M || n >= N) continue;
            const int64_t o = (int64_t)m * ldo + n;
            f32x4 v0 = *(const f32x4 *)(C + row * BN + col);
            if (!ACT_OUT) {
                if (MODE == 1) v0 += rs[i];
                *(f32x4 *)((float *)out + o) = v0;
            } else {
                f32x4 v1 = v0;
                if (W == 8) v1 = *(const f32x4 *)(C + row * BN + col + 4);
                if (ln.stats) {  // LN(x) W^T + b = rstd * (x W'^T - mu c) + d
                    const f32x2 mr = rowtab[row];
                    const float mu = mr[0], rstd = mr[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2 cd = coltab[col + e];
                        v0[e] = fmaf(rstd, fmaf(-mu, cd[0], v0[e]), cd[1]);
                    }
                    if (W == 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2 cd = coltab[col + 4 + e];
                            v1[e] = fmaf(rstd, fmaf(-mu, cd[0], v1[e]), cd[1]);
                        }
                    }
                }
                if (MODE == 2) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 a = gelu_erf2(f32x2{v0[e], v0[e + 1]});
                        v0[e] = a[0];
                        v0[e + 1] = a[1];
                        if (W == 8) {
                            const f32x2 b = gelu_erf2(f32x2{v1[e], v1[e + 1]});
                            v1[e] = b[0];
                            v1[e + 1] = b[1];
                        }
                    }
                }
                if (wt)
                    store_act_wt((OE *)nullptr, (char *)out + (int64_t)m * ldo * (int)sizeof(OE), n, v0, v1);
                else
                    store_act((OE *)nullptr, (char *)out + (int64_t)m * ldo * (int)sizeof(OE), n, v0, v1);
            }
        }
    }
};

// x = resid + acc (bias in the accumulator), fp32 in place — and, for the LayerNorm that follows and is folded into ITS
// consumer: the split pairs of x and the row sums (sum x, sum x^2). Eight lanes own a row of the tile (BN / 64 chunks of
// eight columns each: every load and store is 16 bytes wide), reduce their partial sums by shuffles and store them in the
// tile's slots of the row (launch.h: LnFold — one slot per 64 columns, no atomics, nothing to zero). Both outputs are
// written through (sc1, `wt`): 38 MB of dirty lines per launch at ViT-S/16 B = 64 would otherwise be flushed at the kernel
// boundary, in front of the consumer (measured: attn.qkv / mlp.fc1 7-9 us slower per launch with plain stores).
template <class OE>
struct EpiResidStats {
    static constexpr bool ROWTAB = false;
    static constexpr int BN_MULT = 64;  // tile widths this epilogue can serve
    const float *bias;
    const float *resid;
    float *x;
    void *xs;
    float *stats;
    int M, N;
    int wt = 3;  // bit 0: x, bit 1: xs with write-through stores
    // row centring (launch.h): the pairs and the sums are those of x - shift[row], shift[row] = the row's mean at the previous site
    float *shift = nullptr;
    const float *prev_stats = nullptr, *prev_shift = nullptr;
    __device__ __forceinline__ static void st16(void *p, const f32x4 &v, bool through) {
        if (through)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
        else
            *(f32x4 *)p = v;
    }
    template <class Cfg>
    __device__ __forceinline__ void run(const float *C, int m0, int n0, const f32x2 *, const f32x2 *) const {
        constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, NV = BN / 64, RPP = NT / 8;
        static_assert(BN % 64 == 0 && NT % 64 == 0, "eight lanes x eight columns per chunk");
        const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
        // the residual rows of ALL passes are requested before the first pass stores anything: x is updated in place (resid
        // aliases x), so hipcc keeps a later pass's loads behind an earlier pass's stores — one more exposed L2 round trip per pass
        constexpr int NPASS = (BM + RPP - 1) / RPP, NPRE = NPASS <= 2 ? NPASS : 1;
        f32x4 pre[NPRE][NV][2];
#pragma unroll
        for (int ps = 0; ps < NPRE; ++ps) {
            const int64_t mo = (int64_t)min(m0 + ps * RPP + grp, M - 1) * N;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                pre[ps][i][0] = *(const f32x4 *)(resid + mo + n0 + sub * 8 + i * 64);
                pre[ps][i][1] = *(const f32x4 *)(resid + mo + n0 + sub * 8 + i * 64 + 4);
            }
        }
#pragma unroll 2
        for (int r0 = 0; r0 < BM; r0 += RPP) {
            const int row = r0 + grp, m = m0 + row;
            const bool ok = row < BM && m < M;  // no barriers below; the shuffles are executed by every lane
            const int64_t mo = (int64_t)min(m, M - 1) * N;
            f32x4 v[NV][2];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (NPRE == NPASS) {
                    v[i][0] = pre[NPASS == 2 && r0 ? 1 : 0][i][0];
                    v[i][1] = pre[NPASS == 2 && r0 ? 1 : 0][i][1];
                } else {
                    v[i][0] = *(const f32x4 *)(resid + mo + n0 + sub * 8 + i * 64);
                    v[i][1] = *(const f32x4 *)(resid + mo + n0 + sub * 8 + i * 64 + 4);
                }
            }
            float sh = 0.f;
            if (shift) {  // the eight lanes of the row add the previous site's slots (fixed tree: the same bits in every tile)
                const int nslot = N >> 6;
                float ps = 0.f;
                if (prev_stats)
                    for (int i = sub; i < nslot; i += 8) ps += prev_stats[((int64_t)min(m, M - 1) * nslot + i) * 2];
#pragma unroll
                for (int o = 4; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
                sh = (prev_shift ? prev_shift[min(m, M - 1)] : 0.f) + ps / (float)N;
                if (ok && n0 == 0 && sub == 0) shift[m] = sh;
            }
            float s1 = 0.f, s2 = 0.f;
            char *rowp = (char *)xs + mo * (int)sizeof(OE);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int n = n0 + sub * 8 + i * 64;
                const float *cp = C + min(row, BM - 1) * BN + sub * 8 + i * 64;
                v[i][0] += *(const f32x4 *)cp;
                v[i][1] += *(const f32x4 *)(cp + 4);
                if (ok) {
                    st16(x + mo + n, v[i][0], wt & 1);
                    st16(x + mo + n + 4, v[i][1], wt & 1);
                }
                v[i][0] -= sh;  // from here on the centred row: pairs and sums
                v[i][1] -= sh;
                if (ok) {
                    if constexpr (Elem<OE>::MODE == 2) {
                        bf16x8 hi, lo;
                        split8(v[i][0], v[i][1], hi, lo);
                        st16(rowp + sp_off(n), __builtin_bit_cast(f32x4, hi), wt & 2);
                        st16(rowp + sp_off(n) + 64, __builtin_bit_cast(f32x4, lo), wt & 2);
                    } else if constexpr (Elem<OE>::MODE == 1) {
                        *(f32x4 *)(rowp + n * 4) = v[i][0];
                        *(f32x4 *)(rowp + n * 4 + 16) = v[i][1];
                    } else {
                        *(bf16x8 *)(rowp + n * 2) = cvt8(v[i][0], v[i][1]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1 += v[i][0][e] + v[i][1][e];
                    s2 = fmaf(v[i][0][e], v[i][0][e], fmaf(v[i][1][e], v[i][1][e], s2));
                }
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (ok && sub < NV) {  // the tile's first slot carries its sums, its other slots zeros (launch.h: LnFold)
                const f32x2 val = {sub ? 0.f : s1, sub ? 0.f : s2};
                *((f32x2 *)stats + (int64_t)m * (N >> 6) + (n0 >> 6) + sub) = val;
            }
        }
    }
};

template <class Cfg, class E, bool SWAP, int KSTEPS, class ALoad, class Epi>
static hipError_t launch_gemm_ks(const ALoad &al, const E *w, int64_t ldw, int M, int N, int K, const Epi &epi,
                                 hipStream_t s) {
    auto kern = gemm_kernel<Cfg, E, SWAP, KSTEPS, ALoad, Epi>;
    constexpr int LDS = Cfg::LDS_BYTES + (Epi::ROWTAB ? (Cfg::BM + Cfg::BN) * 8 : 0);  // + the epilogue's row / column tables
    static unsigned long long optin = 0;
    if (hipError_t e = ensure_lds_optin((const void *)kern, LDS, optin); e != hipSuccess) return e;
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN);
    kern<<<dim3(tiles), dim3(Cfg::NT), LDS, s>>>(al, w, ldw, M, N, K, epi);
    return hipGetLastError();
}

// The K extents of ViT-S/B (D and 4D, and the patch embedding) get a compile-time step count, which
// unlocks the two-step prefetch of gemm_mainloop; any other K runs the generic one-step pipeline.
template <class Cfg, class E, bool SWAP, class ALoad, class Epi>
static hipError_t launch_gemm(const ALoad &al, const E *w, int64_t ldw, int M, int N, int K, const Epi &epi,
                              hipStream_t s) {
    if (K % Elem<E>::KROW) return hipErrorInvalidValue;
    switch (K / Elem<E>::KROW) {
        case 2: return launch_gemm_ks<Cfg, E, SWAP, 2>(al, w, ldw, M, N, K, epi, s);  // Swin stage 0 (K = 96 padded)
        case 3: return launch_gemm_ks<Cfg, E, SWAP, 3>(al, w, ldw, M, N, K, epi, s);  // Swin stage 1 (K = 192 bf16, 96 fp32)
        case 4: return launch_gemm_ks<Cfg, E, SWAP, 4>(al, w, ldw, M, N, K, epi, s);
        case 6: return launch_gemm_ks<Cfg, E, SWAP, 6>(al, w, ldw, M, N, K, epi, s);
        case 12: return launch_gemm_ks<Cfg, E, SWAP, 12>(al, w, ldw, M, N, K, epi, s);
        case 24: return launch_gemm_ks<Cfg, E, SWAP, 24>(al, w, ldw, M, N, K, epi, s);
        case 48: return launch_gemm_ks<Cfg, E, SWAP, 48>(al, w, ldw, M, N, K, epi, s);
        case 96: return launch_gemm_ks<Cfg, E, SWAP, 96>(al, w, ldw, M, N, K, epi, s);
        default: break;
    }
    return launch_gemm_ks<Cfg, E, SWAP, 0>(al, w, ldw, M, N, K, epi, s);
}

// ---- LDS-DMA staged variants (gemm_mainloop_dma) ----
template <class Cfg, class E, int KSTEPS, int NSTAGE, class Epi>
static hipError_t launch_gemm_dma_ks(const E *a, int64_t lda, const E *w, int64_t ldw, int M, int N, int K, const Epi &epi,
                                     hipStream_t s) {
    constexpr int LDS = NSTAGE * (Cfg::BM + Cfg::BN) * 128 + (Epi::ROWTAB ? (Cfg::BM + Cfg::BN) * 8 : 0);  // + the epilogue's row / column tables
    static_assert(LDS <= 160 * 1024, "LDS ring exceeds the CU");
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN);
    auto kern = gemm_dma_kernel<Cfg, E, false, KSTEPS, NSTAGE, Epi>;
    static unsigned long long optin = 0;
    if (hipError_t e = ensure_lds_optin((const void *)kern, LDS, optin); e != hipSuccess) return e;
    kern<<<dim3(tiles), dim3(Cfg::NT), LDS, s>>>(a, lda, w, ldw, M, N, K, epi);
    return hipGetLastError();
}

template <class Cfg, class E, int NSTAGE, class Epi>
static hipError_t launch_gemm_dma(const E *a, int64_t lda, const E *w, int64_t ldw, int M, int N, int K, const Epi &epi,
                                  hipStream_t s) {
    if (K % Elem<E>::KROW) return hipErrorInvalidValue;
    switch (K / Elem<E>::KROW) {
        case 6: return launch_gemm_dma_ks<Cfg, E, 6, NSTAGE, Epi>(a, lda, w, ldw, M, N, K, epi, s);
        case 12: return launch_gemm_dma_ks<Cfg, E, 12, NSTAGE, Epi>(a, lda, w, ldw, M, N, K, epi, s);
        case 24: return launch_gemm_dma_ks<Cfg, E, 24, NSTAGE, Epi>(a, lda, w, ldw, M, N, K, epi, s);
        case 48: return launch_gemm_dma_ks<Cfg, E, 48, NSTAGE, Epi>(a, lda, w, ldw, M, N, K, epi, s);
        case 96: return launch_gemm_dma_ks<Cfg, E, 96, NSTAGE, Epi>(a, lda, w, ldw, M, N, K, epi, s);  // ViT-B mlp.fc2
        default: break;
    }
    return launch_gemm_dma_ks<Cfg, E, 0, NSTAGE, Epi>(a, lda, w, ldw, M, N, K, epi, s);
}

typedef GemmCfg<256, 128, 4, 2> Cfg256x128;
// N = 384 outputs (attn.proj, mlp.fc2) at ~12 k rows: 198 tiles, one 8-wave workgroup per CU, wave tile 32 x 96. Per K step
// 40 KiB of operands for 1152 cycles of MFMA per SIMD, against 56 KiB for the 64 x 384 full-row tile
typedef GemmCfg<128, 192, 4, 2> Cfg128x192;
typedef GemmCfg<128, 256, 2, 4> Cfg128x256;
// Swin's channel counts that are not multiples of 128 (96 and 288 = 3 x 96): the same 32 x 96 wave tile, four waves
typedef GemmCfg<128, 96, 4, 1> Cfg128x96;
template <class Epi, class = void>
struct epi_bn_mult {
    static constexpr int v = 1;
};
template <class Epi>
struct epi_bn_mult<Epi, std::void_t<decltype(Epi::BN_MULT)>> {
    static constexpr int v = Epi::BN_MULT;
};

template <int MODE, class E, class Epi>
static hipError_t launch_linear_epi(const E *a, const E *w, const Epi &epi, int M, int N, int K, hipStream_t s) {
    RowLoader<E> al{a, K};
    if constexpr (Elem<E>::MODE == 2) {
#ifdef OCM_DEV
        switch (OCM_KNOB(0)) {  // development: force a variant (tools/microbench_x3.py)
            case -1: goto reg_staged;
            case 1: if (N % 256 == 0) return launch_gemm_dma<Cfg256x256, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 4: if (N % 128 == 0) return launch_gemm_dma<Cfg128x128, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 6: if (N % 128 == 0) return launch_gemm_dma<Cfg256x128, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 7: if (N % 128 == 0) return launch_gemm_dma<Cfg64x128, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 8: if (N % 128 == 0) return launch_gemm_dma<Cfg64x128, E, 3>(a, K, w, K, M, N, K, epi, s); break;
            case 9: if (N % 128 == 0) return launch_gemm_dma<Cfg64x128, E, 4>(a, K, w, K, M, N, K, epi, s); break;
            case 10: if (N % 64 == 0) return launch_gemm_dma<Cfg64x64, E, 4>(a, K, w, K, M, N, K, epi, s); break;
            case 11: if (N == 384) return launch_gemm_dma<Cfg128x192, E, 3>(a, K, w, K, M, N, K, epi, s); break;
            case 12: if (N % 128 == 0) return launch_gemm_dma<Cfg128x128q, E, 2>(a, K, w, K, M, N, K, epi, s); break;  // 8 waves
            case 13: if (N % 128 == 0) return launch_gemm_dma<Cfg128x128q16, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 14: if (N % 128 == 0) return launch_gemm_dma<Cfg128x128m16, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 19: if (N % 256 == 0 && N >= 1024 && M >= 16384) return launch_gemm_dma<Cfg256x256m16, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            case 23: if (N % 128 == 0 && M >= 4096) return launch_gemm_dma<Cfg160x128q16, E, 2>(a, K, w, K, M, N, K, epi, s); break;
            default: break;
        }
#endif
        // Split-bf16 operands, >= 512 tiles of 128x128 (fc1): LDS-DMA staging, two workgroups per CU. Inside the forward
        // (ViT-S/16, B = 64, same box, alternating runs) fc1 62 -> 57 us; the 64x128 shapes (proj, fc2) and the qkv
        // projection measure the same either way (their stand-alone gains of 10 % do not survive cold operands), so
        // they stay on the register-staged loop.
        // Few rows (the reference's one-tile-per-call loops, M = 197 .. 785): everything is L2-resident and a launch is a
        // handful of workgroups, so the LDS-DMA loop's shorter prologue shows (stand-alone, M = 197: fc1 9.4 -> 7.9 us,
        // fc2 23.1 -> 16.7 us on 64x64 tiles with a 4-deep ring, proj 8.3 -> 7.6 us)
        if (M <= OCM_SMALLM_ROWS && OCM_KNOB(0) == 0) {
            if (K >= 1024 && N % 64 == 0) return launch_gemm_dma<Cfg64x64, E, 4>(a, K, w, K, M, N, K, epi, s);
            if (N % 128 == 0) return launch_gemm_dma<Cfg64x128w, E, OCM_SMALLM_STAGES>(a, K, w, K, M, N, K, epi, s);
        }
        // mlp.fc1 at K = 384 and tens of thousands of rows (the 4096^2 slab sweep: 48 k rows; Swin-T stage 2 at batch 256: 50 k):
        // the same 256 x 256 tile although its K loop is only twelve steps (round 4, knob 0 = 19 against the 128 x 128 tile,
        // alternating on one box: slab sweep 534.6 -> 530.6 ms, Swin-T 10.35 -> 10.27 ms; attn.qkv LOSES on it: 534.6 -> 558 ms)
        if (K == 384 && N % 256 == 0 && N >= 1024 && M >= 16384 && OCM_KNOB(0) != 20)
            return launch_gemm_dma<Cfg256x256m16, E, 2>(a, K, w, K, M, N, K, epi, s);
        if (big_tiles_pay(M, N, K))  // ViT-B at 384^2: 256x256 tiles, one 8-wave workgroup per CU (fc1 1020 -> 944 us), on
            // v_mfma_f32_16x16x32_bf16 since round 4 (B = 128, alternating runs: fc1 965 -> 870 us, fc2 903 -> 820, proj 281 -> 256)
            return launch_gemm_dma<Cfg256x256m16, E, 2>(a, K, w, K, M, N, K, epi, s);
        // Narrow outputs (attn.proj / mlp.fc2 of ViT-S: N = 384) with too few rows for 512 tiles of 128 x 128: 128 x 192
        // tiles, one 8-wave workgroup per CU on a three-stage LDS-DMA ring. Per K step 40 KiB of operands for 1152 cycles
        // of MFMA per SIMD (56 KiB for the 64 x 384 full-row tile, 64 KiB for two 64 x 128 tiles): ViT-S/16 at B = 64,
        // in the forward, mlp.fc2 64 -> 49 us, attn.proj 28.5 -> 21.5 us against the full-row GEMM + LayerNorm kernels
        // Past 512 tiles of 128 x 128 the choice goes by how full the last round is: at 48 405 rows (the 4096^2 slab sweep) 1137
        // tiles of 128 x 128 are 2.2 rounds of 512 slots, 758 of 128 x 192 are 2.96 rounds of 256 (sweep 557.7 -> 551.3 ms); at
        // 50 176 rows (Swin-T stage 2) both shapes fill 77 % of their last round and the two-per-CU tile wins (183 against 215 us).
        if (N % 192 == 0 && N / 192 <= 2 && M >= 4096) {
            const long rb = (M + 127) / 128, t192 = rb * (N / 192), t128 = rb * ((N + 127) / 128);
            const double e192 = (double)t192 / (256.0 * ((t192 + 255) / 256)), e128 = (double)t128 / (512.0 * ((t128 + 511) / 512));
            if (t128 < 512 || e192 > 1.1 * e128) return launch_gemm_dma<Cfg128x192, E, 3>(a, K, w, K, M, N, K, epi, s);
        }
        if (const long t128s = (long)((M + 127) / 128) * (N / 128); N % 128 == 0 && t128s >= 512) {
            // wide outputs (mlp.fc1): the same tile on eight wavefronts, four per SIMD with two workgroups per CU (58.0 -> 56.2 us
            // in the forward on one box, 53.0 -> 51.8 on another; the N = 384 layers lose on it: fc2 56 -> 62, proj 26 -> 27, and
            // so does mlp.fc1 at 48 k rows: slab sweep 530 -> 533 ms)
            // (round 4, on the 16 x 16 MFMA shape: at 48 k rows too — slab sweep 541.8 -> 538.7 ms against the four-wave tile, which
            // on the 32 x 32 shape had been the faster one there; knob 0 = 17: the four-wave tile on the 16 x 16 shape, 542.9 ms)
            // (round 4: narrower outputs that reach this branch — Swin-T stages 2 - 3, N = 384 / 768 at 50 k / 12 k rows — also
            // run faster on it than on the four-wave 32 x 32-shape tile: Swin-T at batch 256 10.35 -> 10.17 ms, knob 0 = 13;
            // knob 0 = 21 keeps the four-wave tile for them)
            if constexpr (MODE == 2 || MODE == 3) {  // (the activation-output epilogues: any tile height)
                // 160-row tiles when they fill the two-per-CU slots better: rounds of 512 workgroups, last one counted by its fill
                const long t160 = (long)((M + 159) / 160) * (N / 128);
                const double e128 = (double)t128s / (512.0 * ((t128s + 511) / 512)), e160 = (double)t160 / (512.0 * ((t160 + 511) / 512));
                if (N >= 1024 && K == 384 && e160 > 1.1 * e128 && OCM_KNOB(0) != 22)
                    return launch_gemm_dma<Cfg160x128q16, E, 2>(a, K, w, K, M, N, K, epi, s);
            }
            if ((N >= 1024 || OCM_KNOB(0) != 21) && OCM_KNOB(0) != 17) return launch_gemm_dma<Cfg128x128q16, E, 2>(a, K, w, K, M, N, K, epi, s);
            if (OCM_KNOB(0) == 17) return launch_gemm_dma<Cfg128x128m16, E, 2>(a, K, w, K, M, N, K, epi, s);
            return launch_gemm_dma<Cfg128x128, E, 2>(a, K, w, K, M, N, K, epi, s);
        }
        // Swin-T's narrow stages (N = 96, 192, 288, 576 at 2e5 .. 8e5 rows): tiles that divide N exactly on the LDS-DMA
        // loop instead of 64 x 64 register-staged tiles with a ragged last column (these GEMMs are bound by the 4-byte
        // activations they stream, not by the matrix pipe)
        if (N % 128 != 0 && M >= 4096) {
            if (N % 192 == 0) return launch_gemm_dma<Cfg128x192, E, 3>(a, K, w, K, M, N, K, epi, s);
            if constexpr (Cfg128x96::BN % epi_bn_mult<Epi>::v == 0)
                if (N % 96 == 0) return launch_gemm_dma<Cfg128x96, E, 2>(a, K, w, K, M, N, K, epi, s);
        }
    }
#ifdef OCM_DEV
    if constexpr (Elem<E>::MODE == 0) {  // development A/B: the LDS-DMA loop on single-bf16 operands (knob 0 = 4 / 7)
        if (OCM_KNOB(0) == 4 && N % 128 == 0) return launch_gemm_dma<Cfg128x128, E, 2>(a, K, w, K, M, N, K, epi, s);
        if (OCM_KNOB(0) == 7 && N % 128 == 0) return launch_gemm_dma<Cfg64x128, E, 2>(a, K, w, K, M, N, K, epi, s);
    }
#endif
reg_staged:
    if constexpr (Elem<E>::MODE == 0)
        if (big_tiles_pay(M, N, K)) return launch_gemm<Cfg256x256, E, false>(al, w, K, M, N, K, epi, s);
    // Tile choice: fill >= 2 workgroups per CU (256 CUs) when the problem allows it.
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (N % 128 == 0 && t128 >= 512) return launch_gemm<Cfg128x128, E, false>(al, w, K, M, N, K, epi, s);
    if (N % 128 == 0 && M > 64) return launch_gemm<Cfg64x128, E, false>(al, w, K, M, N, K, epi, s);
    return launch_gemm<Cfg64x64, E, false>(al, w, K, M, N, K, epi, s);
}

// (external linkage: kernels_gemm_inst.hip instantiates the activation-output modes of the split-bf16 type in an object of
// their own, so that the library's longest compile is not twice as long as the others)
template <int MODE, class E>
hipError_t launch_linear_mode(const E *a, const E *w, const float *bias, const float *resid, void *out, int M,
                              int N, int K, hipStream_t s, const LnFold &ln = LnFold()) {
    EpiLinear<MODE, E> epi{bias, resid, out, M, N, N};
    epi.wt = ocm_wt_mask() & 1;
    if (ln.stats) {  // folded LayerNorm: the bias travels in ln.d, the accumulator starts at zero
        epi.ln = ln;
        epi.bias = nullptr;
        if (OCM_KNOB(6) == 1) epi.ln.stats = nullptr;  // development timing probe (wrong results): fold arithmetic off
        if (OCM_KNOB(6) == 2) epi.ln.nslot = 1;         // development timing probe (wrong results): one slot load per row
    }
    return launch_linear_epi<MODE, E>(a, w, epi, M, N, K, s);
}

// Split-K form of the LDS-DMA nn.Linear kernel for few rows and a long contraction (launch.h: StatsOut::part): blockIdx.y is
// the K slice, the accumulator tile goes to part[slice] as plain fp32 (no bias); launch_splitk_finish adds the slices.
template <class Cfg, class E, int KSTEPS, int NSTAGE>
__global__ __launch_bounds__(Cfg::NT) void gemm_dma_splitk_kernel(const E *__restrict__ A, int64_t lda,
                                                                  const E *__restrict__ W, int64_t ldw, int M, int N,
                                                                  int Kslice, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n, slice = blockIdx.y;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    EpiLinear<0, E> epi{nullptr, nullptr, part + (size_t)slice * M * N, M, N, N};
    f32x16 acc[Cfg::TM][Cfg::TN];
    const auto pre = epi_prefetch<Cfg>(epi, m0, n0, M, N);
    gemm_mainloop_dma<Cfg, E, false, KSTEPS, NSTAGE>(A + (size_t)slice * Kslice, lda, W + (size_t)slice * Kslice, ldw, m0, n0, M,
                                                     N, Kslice, smem, acc, (const float *)nullptr);
    run_epilogue<Cfg, false, EpiLinear<0, E>, NSTAGE * (Cfg::BM + Cfg::BN) * 128, decltype(pre), accl_of<Cfg, E>()>(acc, smem, epi, m0, n0, true,
                                                                                                  &pre);
}

template <class E>
static hipError_t launch_linear_splitk(const E *a, const E *w, const float *bias, const float *resid, float *x, int M, int N,
                                       int K, hipStream_t s, const StatsOut &so) {
    typedef Cfg64x64 Cfg;  // (the eight-wave 64 x 128 tile halves the workgroups: 0.628 -> 0.641 ms per one-tile forward)
    constexpr int NSTAGE = 4, KS = 12;  // K / OCM_SPLITK = 384 elements = 12 steps (ViT-S mlp.fc2); other depths: no split
    constexpr int LDS = NSTAGE * (Cfg::BM + Cfg::BN) * 128;
    auto kern = gemm_dma_splitk_kernel<Cfg, E, KS, NSTAGE>;
    static unsigned long long optin = 0;
    if (hipError_t e = ensure_lds_optin((const void *)kern, LDS, optin); e != hipSuccess) return e;
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * (N / Cfg::BN);
    kern<<<dim3(tiles, OCM_SPLITK), dim3(Cfg::NT), LDS, s>>>(a, K, w, K, M, N, K / OCM_SPLITK, so.part);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    return launch_splitk_finish(so.part, OCM_SPLITK, bias, resid, x, so.xs, so.stats, M, N, s, so);
}

// epilogue 4 (internal): x = resid + acc + bias in place, plus split pairs and row sums of x (StatsOut)
template <class E>
hipError_t launch_linear_e(const E *a, const E *w, const float *bias, const float *resid, void *out, int M, int N,
                                  int K, int epilogue, hipStream_t s, const LnFold &ln, const StatsOut &so) {
    switch (epilogue) {
        case 0: return launch_linear_mode<0, E>(a, w, bias, resid, out, M, N, K, s);
        case 1:
            if constexpr (Elem<E>::MODE == 2)
                if (so.part && M <= OCM_SPLITK_MAX_ROWS && N % 64 == 0 && K == OCM_SPLITK * 12 * Elem<E>::KROW && resid &&
                    (!so.stats || so.xs))
                    return launch_linear_splitk<E>(a, w, bias, resid, (float *)out, M, N, K, s, so);
            if (so.stats) {
                if (N % 64) return hipErrorInvalidValue;
                EpiResidStats<E> epi{bias, resid, (float *)out, so.xs, so.stats, M, N};
                epi.shift = so.shift, epi.prev_stats = so.prev_stats, epi.prev_shift = so.prev_shift;
                epi.wt = OCM_KNOB(5) ? OCM_KNOB(5) - 1 : 0;  // development knob 5: 1 + write-through mask. Plain stores ship: x written
                // through costs the producers 3 us per launch and buys the consumers nothing (in-forward A/B)
                return launch_linear_epi<1, E>(a, w, epi, M, N, K, s);
            }
            return launch_linear_mode<1, E>(a, w, bias, resid, out, M, N, K, s);
        case 2: return launch_linear_mode<2, E>(a, w, bias, resid, out, M, N, K, s, ln);
        case 3: return launch_linear_mode<3, E>(a, w, bias, resid, out, M, N, K, s, ln);
    }
    return hipErrorInvalidValue;
}

// nn.Linear with explicit row strides (Swin: activations padded to the GEMM's K step, N not a multiple of the
// tile). N tails are handled by the loaders' row clamps and the epilogue's column guards.
template <int MODE, class E>
static hipError_t launch_linear_ld_mode(const E *a, int64_t lda, const E *w, const float *bias, const float *resid,
                                        void *out, int64_t ldo, int M, int N, int K, hipStream_t s) {
    RowLoader<E> al{a, lda};
    EpiLinear<MODE, E> epi{bias, resid, out, M, N, ldo};
    if (M >= 2048 && N > 64) return launch_gemm<Cfg128x128, E, false>(al, w, K, M, N, K, epi, s);
    if (M > 64 && N > 64) return launch_gemm<Cfg64x128, E, false>(al, w, K, M, N, K, epi, s);
    return launch_gemm<Cfg64x64, E, false>(al, w, K, M, N, K, epi, s);
}

template <class E>
hipError_t launch_linear_ld_e(const E *a, int64_t lda, const E *w, const float *bias, const float *resid, void *out,
                                     int64_t ldo, int M, int N, int K, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case 0: return launch_linear_ld_mode<0, E>(a, lda, w, bias, resid, out, ldo, M, N, K, s);
        case 1: return launch_linear_ld_mode<1, E>(a, lda, w, bias, resid, out, ldo, M, N, K, s);
        case 2: return launch_linear_ld_mode<2, E>(a, lda, w, bias, resid, out, ldo, M, N, K, s);
        case 3: return launch_linear_ld_mode<3, E>(a, lda, w, bias, resid, out, ldo, M, N, K, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// proj / fc2 + residual + the LayerNorm that follows (Block.forward :110-111 then :107 / :111 of the next use)
// ------------------------------------------------------------------------------------------
// A workgroup owns FULL rows (tile 64 x D), so after x = resid + acc its epilogue can normalise the rows it has just
// produced and hand the next GEMM its operand directly: one launch and one read + write of the residual stream less
// per LayerNorm (23 -> 1 LayerNorm launches per ViT-S forward). Statistics are the two-pass fp32 form of
// layernorm_v4_kernel (half a wavefront per row, the row in registers), on exactly the fp32 values written to x.
template <class OE, int D_>
struct EpiResidLN {
    static constexpr bool ROWTAB = false;
    const float *bias;
    const float *resid;
    float *x;           // [M][D] fp32 residual stream out (may alias resid)
    const float *gamma, *beta;
    void *xn;           // [M][D] OE: LayerNorm(x) for the next GEMM
    int M;
    float eps;
    int wt = 0;         // bit 0: xn with write-through stores
    template <class Cfg>
    __device__ __forceinline__ void run(const float *C, int m0, int, const f32x2 *, const f32x2 *) const {
        constexpr int BM = Cfg::BM, NT = Cfg::NT, NV = D_ / 128;
        static_assert(Cfg::BN == D_ && D_ % 128 == 0, "full rows, float4 lanes");
        const int sub = threadIdx.x & 31, half = threadIdx.x >> 5;  // half-wavefront per row
        constexpr int RPP = NT / 32;                                   // rows per pass
        f32x4 g[NV], b[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            g[i] = *(const f32x4 *)(gamma + sub * 4 + i * 128);
            b[i] = *(const f32x4 *)(beta + sub * 4 + i * 128);
        }
#pragma unroll 2
        for (int r0 = 0; r0 < BM; r0 += RPP) {
            const int row = r0 + half, m = m0 + row;
            if (m >= M) continue;  // no barriers below
            f32x4 v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *(const f32x4 *)(resid + (int64_t)m * D_ + sub * 4 + i * 128);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                v[i] += *(const f32x4 *)(C + row * D_ + sub * 4 + i * 128);
                if (wt & 2)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(x + (int64_t)m * D_ + sub * 4 + i * 128), "v"(v[i]) : "memory");
                else
                    *(f32x4 *)(x + (int64_t)m * D_ + sub * 4 + i * 128) = v[i];
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * (1.0f / D_);
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
            const float rstd = 1.0f / sqrtf(q * (1.0f / D_) + eps);
            char *rowp = (char *)xn + (int64_t)m * D_ * (int)sizeof(OE);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = sub * 4 + i * 128;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[i][e] + b[i][e];
                if constexpr (Elem<OE>::MODE == 0) {
                    bf16x4 ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ob[e] = (bf16)o[e];
                    *(bf16x4 *)(rowp + c * 2) = ob;
                } else if constexpr (Elem<OE>::MODE == 1) {
                    *(f32x4 *)(rowp + c * 4) = o;
                } else {
                    bf16x4 hi, lo;
                    split4(o, hi, lo);
                    if (wt & 1) {
                        const f32x2 dh = __builtin_bit_cast(f32x2, hi), dl = __builtin_bit_cast(f32x2, lo);
                        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(rowp + sp_off(c)), "v"(dh) : "memory");
                        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(rowp + sp_off(c) + 64), "v"(dl) : "memory");
                    } else {
                        *(bf16x4 *)(rowp + sp_off(c)) = hi;
                        *(bf16x4 *)(rowp + sp_off(c) + 64) = lo;
                    }
                }
            }
        }
    }
};

// One full-row GEMM + residual + LayerNorm launch on tile configuration Cfg. `loop`: 0 = register-staged main loop,
// 2 = two-stage LDS-DMA ring.
template <class Cfg, class E, int D_>
static hipError_t launch_resid_ln_cfg(int loop, const E *a, const E *w, const EpiResidLN<E, D_> &epi, int M, int K,
                                      hipStream_t s) {
    if (K % Elem<E>::KROW) return hipErrorInvalidValue;
    if (loop == 2) {
        switch (K / Elem<E>::KROW) {
            case 12: return launch_gemm_dma_ks<Cfg, E, 12, 2>(a, K, w, K, M, D_, K, epi, s);
            case 48: return launch_gemm_dma_ks<Cfg, E, 48, 2>(a, K, w, K, M, D_, K, epi, s);
            default: return launch_gemm_dma_ks<Cfg, E, 0, 2>(a, K, w, K, M, D_, K, epi, s);
        }
    }
    RowLoader<E> al{a, K};
    return launch_gemm<Cfg, E, false>(al, w, K, M, D_, K, epi, s);
}

template <class E, int D_>
static hipError_t launch_resid_ln_d(const E *a, const E *w, const float *bias, const float *resid, float *x,
                                    const float *gamma, const float *beta, void *xn, int M, int K, float eps,
                                    hipStream_t s) {
    typedef GemmCfg<64, D_, 2, 4> Cfg8;  // 8 waves, wave tile 32 x D/4
    EpiResidLN<E, D_> epi{bias, resid, x, gamma, beta, xn, M, eps};
    epi.wt = ((ocm_wt_mask() >> 2) & 1) | ((ocm_wt_mask() >> 4) & 2);  // mask 4: xn, mask 32: x
#ifdef OCM_DEV
    if constexpr (Elem<E>::MODE == 2) {
        // development A/B (knob 4): 2 = the 8-wave tile on the two-stage LDS-DMA loop
        if (OCM_KNOB(4) == 2) return launch_resid_ln_cfg<Cfg8, E, D_>(2, a, w, epi, M, K, s);
    }
#endif
    // One 8-wave workgroup per CU on the register-staged loop (two-step prefetch of both operands)
    return launch_resid_ln_cfg<Cfg8, E, D_>(0, a, w, epi, M, K, s);
}

template <class E>
hipError_t launch_resid_ln_e(const E *a, const E *w, const float *bias, const float *resid, float *x,
                                    const float *gamma, const float *beta, void *xn, int M, int D, int K, float eps,
                                    hipStream_t s) {
    switch (D) {
        case 128: return launch_resid_ln_d<E, 128>(a, w, bias, resid, x, gamma, beta, xn, M, K, eps, s);
        case 256: return launch_resid_ln_d<E, 256>(a, w, bias, resid, x, gamma, beta, xn, M, K, eps, s);
        case 384: return launch_resid_ln_d<E, 384>(a, w, bias, resid, x, gamma, beta, xn, M, K, eps, s);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// qkv projection -> head-major q, k and key-contiguous V^T
// ------------------------------------------------------------------------------------------
// Wqkv rows are ordered q(h0..hH-1), k(...), v(...), each head's 64 rows contiguous (:80).
// Column tiles inside [0, 2D) produce q/k rows  dst[(b*H + head)][t][d]          (d contiguous);
// column tiles inside [2D, 3D) run the main loop with the MFMA operands swapped, so the
// accumulator is transposed (lane = token) and V is written as V^T  vt[(b*H + head)][d][t]
// with the token index contiguous — the layout the P·V MFMA consumes — at full store width.
template <class E>
struct EpiQK {
    const float *bias;
    E *q, *k;      // head-major operand copies [B*H][n_pad][hd] (hd 64, or 128 as split pairs), or nullptr
    float *qkv32;  // optional (3,B,H,N,hd) fp32, or nullptr
    int M, ntok, npad, H, D, B, hd;
    int wt = 0;  // write-through stores for q / k
    LnFold ln;   // norm1 folded into this projection (the accumulator then starts at 0)
    static constexpr bool ROWTAB = true;
    __device__ __forceinline__ void row_raw(int m, f32x2 (&raw)[EPI_MAXS]) const { ln_row_raw(ln, m, raw); }
    __device__ __forceinline__ f32x2 row_final(const f32x2 (&raw)[EPI_MAXS]) const { return ln_row_final(ln, raw); }
    __device__ __forceinline__ f32x2 col_entry(int n) const { return ln_col_entry(ln, n); }
    // C is [BM][BN] (rows = tokens). One lane moves 8 consecutive head-dim columns of one token. A lane keeps its
    // column chunk for the whole tile and walks the rows in constant steps, so which / head / d are computed once
    // and (image b, token t) advance incrementally: no integer division per chunk.
    template <class Cfg>
    __device__ __forceinline__ void run(const float *C, int m0, int n0, const f32x2 *rowtab, const f32x2 *coltab) const {
        constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, CPR = BN / 8;
        static_assert(NT % CPR == 0 && BM % (NT / CPR) == 0, "a lane keeps one column chunk");
        constexpr int RSTEP = NT / CPR, ITERS = BM / RSTEP;
        const int col = (threadIdx.x % CPR) * 8, row0 = threadIdx.x / CPR;
        const int n = n0 + col;
        const int which = n / D, rem = n - which * D;
        const int head = hd == 64 ? rem >> 6 : rem / hd, d = rem - head * hd;
        int m = m0 + row0;
        int b = m / ntok, t = m - b * ntok;
        E *base = which ? k : q;
        f32x2 cd[8];  // (c, d) of this lane's eight columns
#pragma unroll
        for (int e = 0; e < 8; ++e) cd[e] = ln.stats ? coltab[col + e] : f32x2{0.f, 0.f};
#pragma unroll 4
        for (int i = 0; i < ITERS; ++i) {
            if (m >= M) break;
            const int row = row0 + i * RSTEP;
            f32x4 v0 = *(const f32x4 *)(C + row * BN + col);
            f32x4 v1 = *(const f32x4 *)(C + row * BN + col + 4);
            if (ln.stats) {
                const f32x2 mr = rowtab[row];
                const float mu = mr[0], rstd = mr[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = fmaf(rstd, fmaf(-mu, cd[e][0], v0[e]), cd[e][1]);
                    v1[e] = fmaf(rstd, fmaf(-mu, cd[4 + e][0], v1[e]), cd[4 + e][1]);
                }
            }
            if (base) {
                char *rowp = (char *)base + ((int64_t)(b * H + head) * npad + t) * hd * (int)sizeof(E);
                if (Elem<E>::EPW == 8) {
                    if (wt)
                        store_act_wt((E *)nullptr, rowp, d, v0, v1);
                    else
                        store_act((E *)nullptr, rowp, d, v0, v1);
                } else {
                    store_act((E *)nullptr, rowp, d, v0, v0);
                    store_act((E *)nullptr, rowp, d + 4, v1, v1);
                }
            }
            if (qkv32) {
                float *o = qkv32 + ((((int64_t)which * B + b) * H + head) * ntok + t) * hd + d;
                *(f32x4 *)o = v0;
                *(f32x4 *)(o + 4) = v1;
            }
            m += RSTEP;
            t += RSTEP;
            while (t >= ntok) {
                t -= ntok;
                ++b;
            }
        }
    }
};

template <class E>
struct EpiVt {
    const float *bias;
    E *vt;  // key-contiguous V^T [B*H][hd][n_pad] (hd 64, or 128 as split pairs); a null vt with a non-null qkv32 still computes the V third
    float *qkv32;
    int M, ntok, npad, H, D, B, hd;
    bool want_v;  // compute the V third at all
    LnFold ln;
    static constexpr bool ROWTAB = true;
    __device__ __forceinline__ void row_raw(int m, f32x2 (&raw)[EPI_MAXS]) const { ln_row_raw(ln, m, raw); }
    __device__ __forceinline__ f32x2 row_final(const f32x2 (&raw)[EPI_MAXS]) const { return ln_row_final(ln, raw); }
    __device__ __forceinline__ f32x2 col_entry(int n) const { return ln_col_entry(ln, n); }
    // C is the TRANSPOSED tile [BN][BM] (rows = features n, columns = tokens m). Consecutive lanes
    // take consecutive tokens of one feature row, so each store instruction writes contiguous runs
    // of V^T (vt[(b*H+head)][d][t], t contiguous).
    template <class Cfg>
    __device__ __forceinline__ void run(const float *C, int m0, int n0, const f32x2 *rowtab, const f32x2 *coltab) const {
        constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
        const int rem0 = n0 - 2 * D;
        // Split pairs, one tile per call (and any tile that lies inside one image at an aligned offset): a lane owns EIGHT
        // consecutive tokens of one feature row — two 16-byte stores (hi, lo) instead of sixteen 2-byte ones; the ragged last
        // piece of the sequence goes element by element.
        if constexpr (Elem<E>::MODE == 2 && NT >= BM / 8) {
            // chosen per tile (workgroup-uniform): all its rows in one image, starting at a multiple of eight tokens. Other
            // tiles keep the token-per-lane form below, whose element stores are contiguous across the wavefront.
            const int b0 = m0 / ntok, t0 = m0 - b0 * ntok;
            if (vt && (t0 & 7) == 0 && t0 + min(BM, M - m0) <= ntok) {
                constexpr int GPR = BM / 8;  // token groups per feature row
                for (int q = threadIdx.x; q < BN * GPR; q += NT) {
                    const int row = q / GPR, g = q - row * GPR, m = m0 + g * 8;
                    if (m >= M) continue;
                    const int b = m / ntok, t = m - b * ntok;
                    const int rem = rem0 + row;
                    const int head = hd == 64 ? rem >> 6 : rem / hd, d = rem - head * hd;
                    f32x4 v0 = *(const f32x4 *)(C + row * BM + g * 8), v1 = *(const f32x4 *)(C + row * BM + g * 8 + 4);
                    if (ln.stats) {
                        const f32x2 cd = coltab[row];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x2 r0 = rowtab[g * 8 + e], r1 = rowtab[g * 8 + 4 + e];
                            v0[e] = fmaf(r0[1], fmaf(-r0[0], cd[0], v0[e]), cd[1]);
                            v1[e] = fmaf(r1[1], fmaf(-r1[0], cd[0], v1[e]), cd[1]);
                        }
                    }
                    char *rowp = (char *)vt + ((int64_t)(b * H + head) * hd + d) * npad * 4;
                    if ((t & 7) == 0 && t + 8 <= ntok && m + 8 <= M) {
                        bf16x8 hi, lo;
                        split8(v0, v1, hi, lo);
                        *(bf16x8 *)(rowp + sp_off(t)) = hi;
                        *(bf16x8 *)(rowp + sp_off(t) + 64) = lo;
                        if (qkv32) {
                            float *o = qkv32 + ((((int64_t)2 * B + b) * H + head) * ntok + t) * hd + d;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[(int64_t)e * hd] = v0[e];
                                o[(int64_t)(4 + e) * hd] = v1[e];
                            }
                        }
                    } else {
                        int bb = b, tt = t;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (m + e < M) {
                                const float v = e < 4 ? v0[e & 3] : v1[e & 3];
                                store_act1((E *)nullptr, (char *)vt + ((int64_t)(bb * H + head) * hd + d) * npad * 4, tt, v);
                                if (qkv32) qkv32[((((int64_t)2 * B + bb) * H + head) * ntok + tt) * hd + d] = v;
                            }
                            if (++tt == ntok) {
                                tt = 0;
                                ++bb;
                            }
                        }
                    }
                }
                return;
            }
        }
        constexpr int RS = NT >= BM ? NT / BM : 1;  // feature rows handled per sweep
        static_assert(NT >= BM, "one lane per token column");
        if (threadIdx.x >= RS * BM) return;
        const int col = threadIdx.x % BM, m = m0 + col;
        if (m >= M) return;
        const int b = m / ntok, t = m - b * ntok;
        const f32x2 mr = rowtab[col];  // (0, 1) without a folded LayerNorm
        const float mu = mr[0], rstd = mr[1];
#pragma unroll 4
        for (int row = threadIdx.x / BM; row < BN; row += RS) {
            const int rem = rem0 + row;
            const int head = hd == 64 ? rem >> 6 : rem / hd, d = rem - head * hd;  // no integer division on the hot path
            float v = C[row * BM + col];
            if (ln.stats) {
                const f32x2 cd = coltab[row];
                v = fmaf(rstd, fmaf(-mu, cd[0], v), cd[1]);
            }
            if (vt) store_act1((E *)nullptr, (char *)vt + ((int64_t)(b * H + head) * hd + d) * npad * (int)sizeof(E), t, v);
            if (qkv32) qkv32[((((int64_t)2 * B + b) * H + head) * ntok + t) * hd + d] = v;
        }
    }
};

template <class Cfg, class E, int KSTEPS>
__global__ __launch_bounds__(Cfg::NT) void qkv_kernel(RowLoader<E> al, const E *__restrict__ W, int M, int D,
                                                      EpiQK<E> eqk, EpiVt<E> ev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = 3 * D, K = D;
    // D % BN == 0 is checked by the launcher; without a V^T destination (a block that stops after its
    // attention probabilities) the V third of the projection is not computed at all
    const int tiles_n = (ev.want_v ? N : 2 * D) / Cfg::BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    f32x16 acc[Cfg::TM][Cfg::TN];
    const auto pre = epi_prefetch<Cfg>(eqk, m0, n0, M, N);
    if (n0 < 2 * D) {  // workgroup-uniform
        gemm_mainloop<Cfg, E, false, KSTEPS>(al, W, K, m0, n0, M, N, K, smem, acc, eqk.bias);
        run_epilogue<Cfg, false, EpiQK<E>, Cfg::LDS_BYTES, decltype(pre), accl_of<Cfg, E>()>(acc, smem, eqk, m0, n0, true, &pre);
    } else {
        gemm_mainloop<Cfg, E, true, KSTEPS>(al, W, K, m0, n0, M, N, K, smem, acc, ev.bias);
        run_epilogue<Cfg, true, EpiVt<E>, Cfg::LDS_BYTES, decltype(pre), accl_of<Cfg, E>()>(acc, smem, ev, m0, n0, true, &pre);
    }
}

// LDS-DMA staged variant (gemm_mainloop_dma); dynamic LDS = NSTAGE * (BM + BN) * 128
template <class Cfg, class E, int KSTEPS, int NSTAGE>
__global__ __launch_bounds__(Cfg::NT) void qkv_dma_kernel(const E *__restrict__ A, const E *__restrict__ W, int M, int D,
                                                          EpiQK<E> eqk, EpiVt<E> ev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = 3 * D, K = D;
    const int tiles_n = (ev.want_v ? N : 2 * D) / Cfg::BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    f32x16 acc[Cfg::TM][Cfg::TN];
    const auto pre = epi_prefetch<Cfg>(eqk, m0, n0, M, N);
    constexpr int RING = NSTAGE * (Cfg::BM + Cfg::BN) * 128;
    if (n0 < 2 * D) {  // workgroup-uniform
        gemm_mainloop_dma<Cfg, E, false, KSTEPS, NSTAGE>(A, K, W, K, m0, n0, M, N, K, smem, acc, eqk.bias);
        run_epilogue<Cfg, false, EpiQK<E>, RING, decltype(pre), accl_of<Cfg, E>()>(acc, smem, eqk, m0, n0, true, &pre);
    } else {
        gemm_mainloop_dma<Cfg, E, true, KSTEPS, NSTAGE>(A, K, W, K, m0, n0, M, N, K, smem, acc, ev.bias);
        run_epilogue<Cfg, true, EpiVt<E>, RING, decltype(pre), accl_of<Cfg, E>()>(acc, smem, ev, m0, n0, true, &pre);
    }
}

template <class Cfg, class E, int KSTEPS, int NSTAGE>
static hipError_t launch_qkv_dma_ks(const E *a, const E *w, int M, int D, const EpiQK<E> &eqk, const EpiVt<E> &ev,
                                    hipStream_t s) {
    auto kern = qkv_dma_kernel<Cfg, E, KSTEPS, NSTAGE>;
    constexpr int LDS = NSTAGE * (Cfg::BM + Cfg::BN) * 128 + (Cfg::BM + Cfg::BN) * 8;  // + the epilogues' row / column tables
    static unsigned long long optin = 0;
    if (hipError_t e = ensure_lds_optin((const void *)kern, LDS, optin); e != hipSuccess) return e;
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((ev.want_v ? 3 : 2) * D / Cfg::BN);
    kern<<<dim3(tiles), dim3(Cfg::NT), LDS, s>>>(a, w, M, D, eqk, ev);
    return hipGetLastError();
}

template <class Cfg, class E, int NSTAGE>
static hipError_t launch_qkv_dma(const E *a, const E *w, int M, int D, const EpiQK<E> &eqk, const EpiVt<E> &ev,
                                 hipStream_t s) {
    switch (D / Elem<E>::KROW) {
        case 12: return launch_qkv_dma_ks<Cfg, E, 12, NSTAGE>(a, w, M, D, eqk, ev, s);
        case 24: return launch_qkv_dma_ks<Cfg, E, 24, NSTAGE>(a, w, M, D, eqk, ev, s);
        default: break;
    }
    return launch_qkv_dma_ks<Cfg, E, 0, NSTAGE>(a, w, M, D, eqk, ev, s);
}

template <class Cfg, class E, int KSTEPS>
static hipError_t launch_qkv_ks(const RowLoader<E> &al, const E *w, int M, int D, const EpiQK<E> &eqk,
                                const EpiVt<E> &ev, hipStream_t s) {
    auto kern = qkv_kernel<Cfg, E, KSTEPS>;
    constexpr int LDS = Cfg::LDS_BYTES + (Cfg::BM + Cfg::BN) * 8;  // + the epilogues' row / column tables
    static unsigned long long optin = 0;
    if (hipError_t e = ensure_lds_optin((const void *)kern, LDS, optin); e != hipSuccess) return e;
    const int tiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((ev.want_v ? 3 : 2) * D / Cfg::BN);
    kern<<<dim3(tiles), dim3(Cfg::NT), LDS, s>>>(al, w, M, D, eqk, ev);
    return hipGetLastError();
}

template <class Cfg, class E>
static hipError_t launch_qkv_cfg(const RowLoader<E> &al, const E *w, int M, int D, const EpiQK<E> &eqk,
                                 const EpiVt<E> &ev, hipStream_t s) {
    switch (D / Elem<E>::KROW) {
        case 6: return launch_qkv_ks<Cfg, E, 6>(al, w, M, D, eqk, ev, s);
        case 12: return launch_qkv_ks<Cfg, E, 12>(al, w, M, D, eqk, ev, s);
        case 24: return launch_qkv_ks<Cfg, E, 24>(al, w, M, D, eqk, ev, s);
        default: break;
    }
    return launch_qkv_ks<Cfg, E, 0>(al, w, M, D, eqk, ev, s);
}

template <class E>
hipError_t launch_qkv_e(const E *a, const E *w, const float *bias, E *q, E *k, E *vt, float *qkv_f32, int batch,
                               int n_tokens, int n_pad, int heads, int head_dim, bool want_v, hipStream_t s,
                               const LnFold &ln) {
    const int D = heads * head_dim, M = batch * n_tokens;
    // operand copies exist for 64-channel heads, and for 128-channel heads as split pairs
    if (head_dim != 64 && head_dim != 128 && (q || k || vt)) return hipErrorInvalidValue;
    if (head_dim % 8) return hipErrorInvalidValue;                      // a lane's 8 columns stay inside one head
    RowLoader<E> al{a, D};
    EpiQK<E> eqk{bias, q, k, qkv_f32, M, n_tokens, n_pad, heads, D, batch, head_dim};
    eqk.wt = (ocm_wt_mask() >> 1) & 1;
    EpiVt<E> ev{bias, vt, qkv_f32, M, n_tokens, n_pad, heads, D, batch, head_dim, want_v};
    if (ln.stats) {  // folded norm1: the bias travels in ln.d, the accumulators start at zero
        eqk.ln = ev.ln = ln;
        eqk.bias = ev.bias = nullptr;
        if (OCM_KNOB(6) == 1) eqk.ln.stats = ev.ln.stats = nullptr;  // development timing probe (wrong results)
        if (OCM_KNOB(6) == 2) eqk.ln.nslot = ev.ln.nslot = 1;
    }
    if constexpr (Elem<E>::MODE == 0)
        if (D % 256 == 0 && big_tiles_pay(M, 3 * D, D)) return launch_qkv_cfg<Cfg256x256, E>(al, w, M, D, eqk, ev, s);
    const long t128 = (long)((M + 127) / 128) * (3 * D / 128);
    if constexpr (Elem<E>::MODE == 2) {
        if (D % 128 == 0 && M > 64) {
#ifdef OCM_DEV
            switch (OCM_KNOB(3)) {  // development: force a variant (-1: the register-staged kernels below)
                case 1: return launch_qkv_dma<Cfg128x128, E, 2>(a, w, M, D, eqk, ev, s);
                case 2: return launch_qkv_dma<Cfg128x128q, E, 2>(a, w, M, D, eqk, ev, s);
                case 3: return launch_qkv_dma<Cfg64x128, E, 2>(a, w, M, D, eqk, ev, s);
                case 4: if (D % 256 == 0) return launch_qkv_dma<Cfg256x256, E, 2>(a, w, M, D, eqk, ev, s); break;
                case 5: return launch_qkv_dma<Cfg128x128q16, E, 2>(a, w, M, D, eqk, ev, s);
                case 6: if (D % 256 == 0) return launch_qkv_dma<Cfg256x256m16, E, 2>(a, w, M, D, eqk, ev, s); break;
                default: break;
            }
#endif
            if (OCM_KNOB(3) == 0) {
                // few rows (one tile per call): the DMA loop's shorter prologue shows (B = 1 forward 1.03 -> 1.01 ms)
                if (M <= OCM_SMALLM_ROWS) return launch_qkv_dma<Cfg64x128w, E, OCM_SMALLM_STAGES>(a, w, M, D, eqk, ev, s);
                // ViT-B sizes: 256 x 256 tiles halve the bytes through L2 (384^2 B = 128: 755 -> 715 us per launch)
                // (on v_mfma_f32_16x16x32_bf16 since round 4: 717 -> 670 us)
                if (D % 256 == 0 && big_tiles_pay(M, 3 * D, D)) return launch_qkv_dma<Cfg256x256m16, E, 2>(a, w, M, D, eqk, ev, s);
                // the 8-wave 128 x 128 tile on the LDS-DMA loop (ViT-S/16 B = 64: 46.6 -> 41.8 us per launch, +2 % end
                // to end; ViT-B/16 384^2 B = 128: 805 -> 759 us; alternating runs on one box). The 4-wave form of the
                // same tile (variant 1) measures like the register-staged kernel.
                // ... on v_mfma_f32_16x16x32_bf16 (43.1 -> 41.6 us; GemmCfg::MF16)
                if (t128 >= 512) return launch_qkv_dma<Cfg128x128q16, E, 2>(a, w, M, D, eqk, ev, s);
            }
        }
    }
    if (D % 128 == 0 && t128 >= 512) return launch_qkv_cfg<Cfg128x128q, E>(al, w, M, D, eqk, ev, s);
    if (D % 128 == 0) return launch_qkv_cfg<Cfg64x128, E>(al, w, M, D, eqk, ev, s);
    return launch_qkv_cfg<Cfg64x64, E>(al, w, M, D, eqk, ev, s);
}

// ------------------------------------------------------------------------------------------
// patch embedding: im2col-free gather from fp32 planes + GEMM + bias + pos-embed
// ------------------------------------------------------------------------------------------
// Row m = b*P + py*wp + px is the p x p patch at (py, px) of tile b (row-major flatten, :131);
// column k = c*p*p + dy*p + dx indexes conv weight (D, C, p, p) flattened (:127). A 16-B LDS
// chunk is 8 (bf16) or 4 (fp32) consecutive dx of one (c, dy): float4 loads from one image row,
// converted to bf16 on the way into LDS in the bf16 path. Consecutive threads walk consecutive
// chunks of a row, so a wave reads whole row segments of neighbouring patches (coalesced along x).
template <class E>
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
    // chunk cc of K step t: bf16 -> k = 64t + 8cc .. +7; fp32 -> k = 32t + 4cc .. +3; split pairs -> the hi (cc < 4)
    // or lo (cc >= 4) halves of k = 32t + 8(cc & 3) .. +7
    __device__ __forceinline__ Raw load(Handle h, int t, int cc) const {
        constexpr int MODE = Elem<E>::MODE;
        const int k = MODE == 0 ? t * 64 + cc * 8 : MODE == 1 ? t * 32 + cc * 4 : t * 32 + (cc & 3) * 8;
        const int c = k / pp, rem = k - c * pp;
        const int dy = rem / p, dx = rem - dy * p;
        const float *ptr = h + (int64_t)c * sc + (int64_t)dy * sy + dx;
        Raw r;
        r.lo = *(const f32x4 *)ptr;
        r.hi = r.lo;
        if (MODE != 1) r.hi = *(const f32x4 *)(ptr + 4);
        return r;
    }
    __device__ __forceinline__ static typename Elem<E>::Chunk finish(const Raw &r, int cc) {
        if constexpr (Elem<E>::MODE == 0) {
            return cvt8(r.lo, r.hi);
        } else if constexpr (Elem<E>::MODE == 1) {
            return r.lo;
        } else {
            bf16x8 hi, lo;
            split8(r.lo, r.hi, hi, lo);
            const bf16x8 sel = cc < 4 ? hi : lo;
            return __builtin_bit_cast(f32x4, sel);
        }
    }
};

// x[b][1 + pi][n] = acc + pos[1 + pi][n]   (prepare_tokens :200-207, patch rows; bias is in acc)
// With a SimMIM mask (model.py:28-33) the patch row is first blended with the mask token:
// acc*(1-w) + mask_token*w, w = mask[b][pi], evaluated in that order in fp32.
struct EpiPatch {
    static constexpr bool ROWTAB = false;
    const float *bias, *pos;
    float *x;
    int M, P, ntok, D;
    const float *mask, *mask_tok;
    StatsOut so;  // split pairs + row sums of the token rows (the first LayerNorm folded into the first qkv projection)
    template <class Cfg>
    __device__ __forceinline__ void run(const float *C, int m0, int n0, const f32x2 *, const f32x2 *) const {
        constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, CPR = BN / 4;
        if (so.stats) return run_stats<Cfg>(C, m0, n0);  // workgroup-uniform
#pragma unroll 4
        for (int q = threadIdx.x; q < BM * CPR; q += NT) {
            const int row = q / CPR, col = (q - row * CPR) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= M || n >= D) continue;
            const int b = m / P, t = m - b * P;
            f32x4 v = *(const f32x4 *)(C + row * BN + col);
            if (mask) {
                const float w = mask[m], omw = 1.0f - w;
                const f32x4 tk = *(const f32x4 *)(mask_tok + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * omw + tk[e] * w;  // exact for the 0/1 masks SimMIM draws
            }
            v += *(const f32x4 *)(pos + (int64_t)(1 + t) * D + n);
            *(f32x4 *)(x + ((int64_t)b * ntok + 1 + t) * D + n) = v;
        }
    }
    // the same rows, also as split pairs, with their sums (the first LayerNorm is folded into the first qkv projection):
    // the CPR lanes that own a row reduce by shuffles, so every lane runs every iteration (rows past M are masked)
    template <class Cfg>
    __device__ __forceinline__ void run_stats(const float *C, int m0, int n0) const {
        constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, CPR = BN / 4;
        static_assert((BM * CPR) % NT == 0 && CPR <= 32 && (CPR & (CPR - 1)) == 0, "CPR lanes of one wave half own a row");
        const int sl = threadIdx.x & (CPR - 1), col = sl * 4, n = n0 + col;  // NT % CPR == 0: a lane keeps its columns
        const bool nok = n < D;
        const int nc = min(n, D - 4);
#pragma unroll 1
        for (int row = threadIdx.x / CPR; row < BM; row += NT / CPR) {
            const int m = m0 + row;
            const bool ok = nok && m < M;
            const int mc = min(m, M - 1);
            const int b = mc / P, t = mc - b * P;
            f32x4 v = *(const f32x4 *)(C + row * BN + col);
            if (mask) {
                const float w = mask[mc], omw = 1.0f - w;
                const f32x4 tk = *(const f32x4 *)(mask_tok + nc);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * omw + tk[e] * w;
            }
            v += *(const f32x4 *)(pos + (int64_t)(1 + t) * D + nc);
            const int64_t xrow = (int64_t)b * ntok + 1 + t;
            const float sh = so.tok_shift ? so.tok_shift[1 + t] : 0.f;  // row centring (launch.h): a per-token constant here
            float s1 = 0.f, s2 = 0.f;
            if (ok) {
                *(f32x4 *)(x + xrow * D + n) = v;
                if (so.shift && n0 == 0 && sl == 0) so.shift[xrow] = sh;
                v -= sh;
                bf16x4 hi, lo;
                split4(v, hi, lo);
                char *g = (char *)so.xs + xrow * D * 4 + sp_off(n);
                *(bf16x4 *)g = hi;
                *(bf16x4 *)(g + 64) = lo;
                s1 = (v[0] + v[1]) + (v[2] + v[3]);
                s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
            }
#pragma unroll
            for (int o = CPR / 2; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (m < M && sl < BN / 64 && n0 + 64 * sl < D) {  // the tile's slots of the row (launch.h: LnFold)
                const f32x2 val = {sl ? 0.f : s1, sl ? 0.f : s2};
                *((f32x2 *)so.stats + xrow * (D >> 6) + (n0 >> 6) + sl) = val;
            }
        }
    }
};

template <class E>
hipError_t launch_patch_e(const PatchArgs &pa, const E *w, const float *bias, const float *pos, float *x, int dim,
                                 hipStream_t s, const StatsOut &so) {
    const int P = pa.hp * pa.wp, M = pa.batch * P, K = pa.chans * pa.p * pa.p;
    PatchLoader<E> al{pa.image, pa.sb, pa.sc, pa.sy, pa.origins, P, pa.wp, pa.p, pa.p * pa.p};
    EpiPatch epi{bias, pos, x, M, P, P + 1, dim, pa.mask, pa.mask_tok, so};
    // one tile per call (a dozen workgroups): eight wavefronts per tile, two per SIMD (see Cfg64x128w)
    if constexpr (Elem<E>::MODE == 2)
        if (dim % 128 == 0 && M > 64 && M <= 1024) return launch_gemm<Cfg64x128w, E, false>(al, w, K, M, dim, K, epi, s);
    if (dim % 128 == 0 && M > 64) return launch_gemm<Cfg64x128, E, false>(al, w, K, M, dim, K, epi, s);
    return launch_gemm<Cfg64x64, E, false>(al, w, K, M, dim, K, epi, s);
}

