// gemm_core.h — MFMA tile GEMM for the ViT projections (gfx950).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        (nn.Linear layout: W is [N][K], K contiguous)
//
// One workgroup = WAVES_M x WAVES_N wavefronts computes a BM x BN tile; K is walked in steps of one
// 128-byte LDS row (64 bf16 or 32 fp32 elements) through a double-buffered, XOR-swizzled LDS image
// (common.h: lds_off). The operand element type selects the matrix instruction:
//   bf16  : v_mfma_f32_32x32x16_bf16   (OCM_PREC_BF16, the fast path)
//   float : v_mfma_f32_32x32x2_f32     (OCM_PREC_FP32: exact fp32 products, 1/16 of the bf16 rate)
// Both accumulate in fp32 and share the C/D register layout, the LDS geometry, the staging code and
// every epilogue.
//
// Staging is register-staged with the issue-early / commit-late split and one LDS-only barrier per
// step. When K is a compile-time constant (KSTEPS > 0) two register slots give a prefetch
// distance of two steps with an unconditional load stream, so hipcc counts its own `vmcnt(N)` waits.
//
// The A operand comes through a loader policy so the same main loop serves
//   - plain row-major activations (RowLoader) and
//   - the im2col-free patch gather from fp32 image planes (PatchLoader, kernels_gemm.hip).
// The accumulator can be produced transposed (SWAP): acc^T has the token index on the lanes, which is
// how the V projection is written key-contiguous (V^T) with coalesced stores.
#pragma once
#include "common.h"

// Development instrumentation (make stamps -> exp_libs/stamps.so, tools/stamps.py): thread 0 of every workgroup
// records the shader clock at the phase boundaries of gemm_kernel. Compiled out unless -DOCM_GEMM_STAMPS.
#ifdef OCM_GEMM_STAMPS
// per-wave timeline of ONE K step (the middle one) of the LDS-DMA loop: [workgroup][wave][point]
__device__ unsigned long long g_wstamps[512 * 16 * 8];
#define WSTAMP(k)                                                                                       \
    do {                                                                                                \
        if (t == nsteps / 2 && (threadIdx.x & 63) == 0 && blockIdx.x < 512)                              \
            g_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6 & 15)) * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
__device__ unsigned long long g_stamps[8192 * 8];
#define STAMP(i)                                                                                        \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(i) ((void)0)
#define WSTAMP(k) ((void)0)
#endif

// MF16_: split-bf16 products on v_mfma_f32_16x16x32_bf16 (four 16 x 16 accumulators per 32 x 32 tile) instead of 32x32x16 — the
// same LDS bytes and the same matrix cycles per product; the chip holds a higher clock on it (MI355X_MICROARCH.md, DVFS
// give-back item 7). Pays where the matrix pipe is the busier resource: the 128 x 128 tiles with two workgroups per CU
// (ViT-S/16 B = 64 in the forward, alternating libraries on one box: mlp.fc1 52.2 -> 49.8 us, attn.qkv 43.1 -> 41.6); the
// one-per-CU 128 x 192 tiles (LDS-port-bound) measure the same and the register-staged patch embedding loses (55 -> 68 us).
template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int MF16_ = 0>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, MF16 = MF16_;
    static constexpr int NT = WAVES_M * WAVES_N * OCM_WAVE;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    // HALF (16 x 16 MFMA shape, LDS-DMA loop only): a wave's LAST 32 x 32 tile is only its upper 16 rows — BM = (TM - 1) * RB + 16 *
    // WAVES_M, the half tiles of the waves forming one band of 16 * WAVES_M rows at the bottom of the block tile. 160-row tiles
    // (TM = 3: two full tiles and a half) exist for row counts whose 128-row tiling leaves a nearly empty last round of workgroups.
    static constexpr int HALF = (WM % 32 == 16) ? 1 : 0;
    static constexpr int TM = (WM + 31) / 32, TN = WN / 32;
    static constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;  // 16-B chunks / thread / K-step
    static constexpr int LDS_BYTES = 2 * (BM + BN) * 128;
    // A wave's 32x32 tiles are INTERLEAVED with the other waves': tile (i, j) of wave (wm, wn) sits at
    // rows (i*WAVES_M + wm)*32, columns (j*WAVES_N + wn)*32 of the block tile, so the i-th (j-th) tiles
    // of all waves together cover a contiguous band of RB (CB) rows (columns) — the unit the epilogue
    // stages through LDS when the whole fp32 tile does not fit.
    static constexpr int RB = 32 * WAVES_M, CB = 32 * WAVES_N;
    static_assert((WM % 32 == 0 || (HALF && MF16_)) && WN % 32 == 0, "wave tile must be a multiple of 32x32 (or end in a 16-row half tile)");
    static_assert(HALF || ((BM * 8) % NT == 0 && (BN * 8) % NT == 0), "staging must divide evenly");
    static_assert(NT % 8 == 0, "a thread keeps one chunk column");
};

// what an epilogue sees of one staged pass: a PBM x PBN fp32 image handled by NT threads
template <int PBM, int PBN, int NT_>
struct PassCfg {
    static constexpr int BM = PBM, BN = PBN, NT = NT_;
};

// 1: 8-wave register-staged split-bf16 main loops run their two half-workgroups in opposite phases (0: A/B builds)
#ifndef OCM_PINGPONG
#define OCM_PINGPONG 1
#endif

// operand element traits: an LDS row (one K step, 128 bytes) holds KROW elements; an epilogue lane writes EPW
// consecutive elements of an activation row. MODE: 0 = bf16, 1 = fp32, 2 = split-bf16 pairs (common.h: sp32).
// In every mode K step t of a row is the 128 bytes at byte offset t * 128 of that row, as eight 16-byte chunks.
template <class E>
struct Elem;
template <>
struct Elem<bf16> {
    typedef bf16x8 Chunk;
    static constexpr int MODE = 0, KROW = 64, EPW = 8;
};
template <>
struct Elem<float> {
    typedef f32x4 Chunk;
    static constexpr int MODE = 1, KROW = 32, EPW = 4;
};
template <>
struct Elem<sp32> {
    typedef f32x4 Chunk;  // 16 opaque bytes: 8 hi or 8 lo halves
    static constexpr int MODE = 2, KROW = 32, EPW = 8;
};

// accumulator layout of element type E's main loop (common.h: acc_rpos / acc_cpos)
template <class Cfg, class E>
constexpr int accl_of() {
    return (Elem<E>::MODE == 2 && Cfg::MF16) ? 1 : 0;
}

// A operand = row-major activations of element type E (lda in elements).
template <class E>
struct RowLoader {
    const E *A;
    int64_t lda;
    typedef const char *Handle;
    typedef typename Elem<E>::Chunk Raw;
    __device__ __forceinline__ Handle row(int m) const { return (const char *)A + (int64_t)m * lda * (int)sizeof(E); }
    // chunk c of K step t
    __device__ __forceinline__ Raw load(Handle h, int t, int c) const {
#if defined(OCM_ABL) && (OCM_ABL == 8 || OCM_ABL == 10)  // ablation 8 (10: both operands): the A operand re-reads its first K step (cache hits)
        t = 0;
#endif
        return *(const Raw *)(h + t * 128 + c * 16);
    }
    __device__ __forceinline__ static Raw finish(const Raw &r, int) { return r; }
};

// One K step of MFMAs on the LDS tiles at Ab / Bb (already offset to the wave's rows).
template <class Cfg, class E, bool SWAP>
__device__ __forceinline__ void mma_step(const char *Ab, const char *Bb, int r, int h,
                                         f32x16 (&acc)[Cfg::TM][Cfg::TN], int wm = 0) {
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    static_assert(!Cfg::HALF || (Elem<E>::MODE == 2 && Cfg::MF16 && !SWAP), "half tiles exist on the 16 x 16 shape only");
    if constexpr (Elem<E>::MODE == 2 && Cfg::MF16) {
        // split-bf16 on 16x16x32: lane (row r & 15 of a 16-row half, k chunk g) reads hi chunk g / lo chunk 4 + g of its row —
        // the whole K step in one instruction per (half, part); twelve MFMAs per 32 x 32 tile and step, four accumulators
        const int r16 = r & 15, g = (r >> 4) + 2 * h;
        bf16x8 ah[TM][2], al[TM][2], bh[TN][2], bl[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ra = 0; ra < 2; ++ra) {
                if (Cfg::HALF && i == TM - 1) {  // Ab = tile rows wm * 32: the half band starts at (TM - 1) * RB, this wave's rows at wm * 16 in it
                    if (ra == 0) {
                        const char *Ah = Ab + ((TM - 1) * Cfg::RB - wm * 16) * 128;
                        ah[i][0] = *(const bf16x8 *)(Ah + lds_off(r16, g));
                        al[i][0] = *(const bf16x8 *)(Ah + lds_off(r16, 4 + g));
                    }
                    continue;
                }
                ah[i][ra] = *(const bf16x8 *)(Ab + i * Cfg::RB * 128 + lds_off(ra * 16 + r16, g));
                al[i][ra] = *(const bf16x8 *)(Ab + i * Cfg::RB * 128 + lds_off(ra * 16 + r16, 4 + g));
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                bh[j][cb] = *(const bf16x8 *)(Bb + j * Cfg::CB * 128 + lds_off(cb * 16 + r16, g));
                bl[j][cb] = *(const bf16x8 *)(Bb + j * Cfg::CB * 128 + lds_off(cb * 16 + r16, 4 + g));
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < ((Cfg::HALF && i == TM - 1) ? 2 : 4); ++q) {
                    // sub-tile q: register-indexed half q >> 1, lane-indexed half q & 1. Normal: registers = rows (A), lanes =
                    // columns (B); SWAP: registers = n (B rows as the first operand), lanes = m
                    f32x4_t c = __builtin_shufflevector(acc[i][j], acc[i][j], 0, 1, 2, 3);
                    if (q == 1) c = __builtin_shufflevector(acc[i][j], acc[i][j], 4, 5, 6, 7);
                    if (q == 2) c = __builtin_shufflevector(acc[i][j], acc[i][j], 8, 9, 10, 11);
                    if (q == 3) c = __builtin_shufflevector(acc[i][j], acc[i][j], 12, 13, 14, 15);
                    c = SWAP ? mfma16x3(bh[j][q >> 1], bl[j][q >> 1], ah[i][q & 1], al[i][q & 1], c)
                             : mfma16x3(ah[i][q >> 1], al[i][q >> 1], bh[j][q & 1], bl[j][q & 1], c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = c[e];
                }
    } else if constexpr (Elem<E>::MODE == 2) {
        // split-bf16: the 128-byte row is [hi k 0..31 | lo k 0..31]; two k16 sub-steps, three MFMAs per product.
        // (hipcc interleaves these reads and MFMAs in groups of three to five reads followed by lgkmcnt(0); pinning a
        // rolling window of reads two blocks ahead of the MFMAs with sched_group_barrier was measured on the 64 x 384
        // tile: no gain — that kernel is paced by LDS bytes, DESIGN.md §3.8 — and the patch embedding 15 % slower.)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *(const bf16x8 *)(Ab + i * Cfg::RB * 128 + lds_off(r, 2 * s + h));
                al[i] = *(const bf16x8 *)(Ab + i * Cfg::RB * 128 + lds_off(r, 4 + 2 * s + h));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *(const bf16x8 *)(Bb + j * Cfg::CB * 128 + lds_off(r, 2 * s + h));
                bl[j] = *(const bf16x8 *)(Bb + j * Cfg::CB * 128 + lds_off(r, 4 + 2 * s + h));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = SWAP ? mfma32x3(bh[j], bl[j], ah[i], al[i], acc[i][j])
                                     : mfma32x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
        }
    } else if constexpr (Elem<E>::MODE == 0) {
        // rows (i*WAVES_M + wm)*32 + r: the swizzle term (row>>1)&7 only depends on r because the tile row
        // offsets are multiples of 32. Lane (r, h) holds k = 16s + 8h .. +7 of row r.
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8 *)(Ab + i * Cfg::RB * 128 + lds_off(r, 2 * s + h));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8 *)(Bb + j * Cfg::CB * 128 + lds_off(r, 2 * s + h));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = SWAP ? mfma32(b[j], a[i], acc[i][j]) : mfma32(a[i], b[j], acc[i][j]);
        }
    } else {
        // v_mfma_f32_32x32x2_f32: lane (r, h) supplies A[r][k_h] and B[k_h][r] for the instruction's two
        // k values. Lane half h reads chunk c + 4h of its row (k = 16h + 4c .. +3): each 16-B LDS read
        // feeds 4 MFMAs whose k pairs are {4c + e, 16 + 4c + e} — any pairing is valid as long as A
        // and B use the same one, which they do (same chunk index on both operands).
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4 *)(Ab + i * Cfg::RB * 128 + lds_off(r, c + 4 * h));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f32x4 *)(Bb + j * Cfg::CB * 128 + lds_off(r, c + 4 * h));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = SWAP ? mfma32f(b[j][e], a[i][e], acc[i][j]) : mfma32f(a[i][e], b[j][e], acc[i][j]);
        }
    }
}

// Main loop. acc[i][j] is the 32x32 tile at rows (i*WAVES_M+wm)*32, cols (j*WAVES_N+wn)*32 of the block tile; with
// SWAP the register/lane roles of that tile are transposed (lane = row m). KSTEPS = K / KROW when K
// is known at compile time (0 = runtime K).
template <class Cfg, class E, bool SWAP, int KSTEPS, class ALoad>
__device__ __forceinline__ void gemm_mainloop(const ALoad &al, const E *__restrict__ W, int64_t ldw, int m0, int n0,
                                              int M, int N, int K, char *smem, f32x16 (&acc)[Cfg::TM][Cfg::TN],
                                              const float *__restrict__ bias) {
    typedef typename Elem<E>::Chunk Chunk;
    static_assert(!Cfg::HALF, "half tiles exist on the LDS-DMA loop only");
    constexpr int KROW = Elem<E>::KROW;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
    constexpr int A_CH = Cfg::A_CH, B_CH = Cfg::B_CH, TM = Cfg::TM, TN = Cfg::TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int r = lane & 31, h = lane >> 5;
    char *As = smem;
    char *Bs = smem + 2 * BM * 128;

    // NT % 8 == 0: a thread keeps chunk column cc = tid & 7 for all of its chunks
    const int cc = tid & 7;
    typename ALoad::Handle a_h[A_CH];
    const char *b_h[B_CH];
    int a_off[A_CH], b_off[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int q = tid + NT * i, row = q >> 3;
        a_h[i] = al.row(min(m0 + row, M - 1));
        a_off[i] = lds_off(row, cc);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int q = tid + NT * i, row = q >> 3;
        b_h[i] = (const char *)W + (int64_t)min(n0 + row, N - 1) * ldw * (int)sizeof(E) + cc * 16;
        b_off[i] = lds_off(row, cc);
    }
    struct Slot {
        typename ALoad::Raw a[A_CH];
        Chunk b[B_CH];
    };
    auto issue = [&](Slot &sl, int t) {  // K step t
#pragma unroll
        for (int i = 0; i < A_CH; ++i) sl.a[i] = al.load(a_h[i], t, cc);
#pragma unroll
#if defined(OCM_ABL) && (OCM_ABL == 9 || OCM_ABL == 10)  // ablation 9 (10: both operands): the W operand re-reads its first K step (cache hits)
        for (int i = 0; i < B_CH; ++i) sl.b[i] = *(const Chunk *)(b_h[i]);
#else
        for (int i = 0; i < B_CH; ++i) sl.b[i] = *(const Chunk *)(b_h[i] + t * 128);
#endif
    };
    auto commit = [&](int buf, const Slot &sl) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *(Chunk *)(As + buf * BM * 128 + a_off[i]) = ALoad::finish(sl.a[i], cc);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *(Chunk *)(Bs + buf * BN * 128 + b_off[i]) = sl.b[i];
    };
    auto compute = [&](int buf) {
        mma_step<Cfg, E, SWAP>(As + buf * BM * 128 + (wm * 32) * 128, Bs + buf * BN * 128 + (wn * 32) * 128, r, h, acc);
    };

    // Accumulators start at the bias (one load per column / register row, issued together with the
    // first operand tile): the epilogues then add nothing and issue no dependent global loads.
    constexpr int ACCL = accl_of<Cfg, E>();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + (j * Cfg::WAVES_N + wn) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {  // normal: the lane-indexed axis is the output column; SWAP: the register-indexed one
            const float bv = bias ? bias[min(nb + (SWAP ? acc_rpos<ACCL>(e, lane) : acc_cpos<ACCL>(e, lane)), N - 1)] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j][e] = bv;
        }
    }

    if constexpr (KSTEPS > 0) {
        // Two register slots (s0: even steps, s1: odd steps): the tile committed to LDS in step t was requested
        // in step t-2 (L2 latency under load is ~1 us, more than one step). The stream is unconditional (indices
        // past the end re-read the last tile into a dead slot) and the loop is fully unrolled with a constant
        // trip count, so hipcc counts its own `s_waitcnt vmcnt(N)` instead of draining to 0.
#if defined(OCM_ABL) && OCM_ABL == 11  // ablation 11: no global loads after the prologue (slots keep their first tiles)
        auto issue2 = [&](Slot &sl, int t) { if (t < 3) issue(sl, min(t, KSTEPS - 1)); };
#else
        auto issue2 = [&](Slot &sl, int t) { issue(sl, min(t, KSTEPS - 1)); };
#endif
        Slot s0, s1;
        issue2(s0, 0);
        issue2(s1, 1);
        commit(0, s0);
        issue2(s0, 2);
        lds_barrier();
#ifdef OCM_GEMM_STAMPS
        STAMP(6);  // first tile in LDS
#endif
        if constexpr (KSTEPS % 2 == 0 && Elem<E>::MODE == 2 && NT == 512 && OCM_PINGPONG) {
            // Eight waves = two per SIMD (waves w and w + 4 share SIMD w): run the two halves of the workgroup in
            // OPPOSITE phases instead of in lock step. Per K step two intervals, a barrier after each:
            //   interval 1: waves 0-3 multiply step t            | waves 4-7 commit their share of step t+1, issue t+3
            //   interval 2: waves 0-3 commit their share of t+1  | waves 4-7 multiply step t
            // so on every SIMD one wave's MFMAs run beside the other wave's waits, ds_writes and load issue, instead of
            // both waves paying the LDS read latency and the wait -> ds_write -> barrier tail at the same time
            // (stamps: 2 108 cycles per step against 1 152 of MFMA issue in lock step). Buffers: step t is read from
            // buffer t & 1 in both intervals while shares of step t+1 land in the other buffer; a buffer is refilled
            // only after the end-of-step barrier that follows its last reader. Same MFMA order per accumulator:
            // bit-identical results.
            // nothing may be scheduled across an interval boundary (hipcc otherwise sinks MFMAs below the barrier,
            // into the partner's MFMA interval)
            auto phase_barrier = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
            if (grp == 0) {
                for (int t = 0; t < KSTEPS; t += 2) {
                    compute(0);  // step t
                    phase_barrier();
                    commit(1, s1);  // own share of step t+1
                    issue2(s1, t + 3);
                    phase_barrier();
                    compute(1);  // step t+1
                    phase_barrier();
                    commit(0, s0);  // own share of step t+2 (a dead duplicate after the last step)
                    issue2(s0, t + 4);
                    phase_barrier();
                }
            } else {
                for (int t = 0; t < KSTEPS; t += 2) {
                    commit(1, s1);
                    issue2(s1, t + 3);
                    phase_barrier();
                    compute(0);
                    phase_barrier();
                    commit(0, s0);
                    issue2(s0, t + 4);
                    phase_barrier();
                    compute(1);
                    phase_barrier();
                }
            }
        } else if constexpr (KSTEPS % 2 == 0) {
            for (int t = 0; t < KSTEPS; t += 2) {
                compute(0);     // step t (even steps live in buffer 0)
                commit(1, s1);  // step t+1
                issue2(s1, t + 3);
                lds_barrier();
                compute(1);     // step t+1
                commit(0, s0);  // step t+2 (a dead duplicate after the last step)
                issue2(s0, t + 4);
                lds_barrier();
            }
        } else {  // small odd step counts (K = 192 bf16, K = 96 fp32): fully unrolled
#pragma unroll
            for (int t = 0; t < KSTEPS; ++t) {
                compute(t & 1);  // step t lives in buffer t & 1
                if (t & 1) {
                    if (t + 1 < KSTEPS) commit(0, s0);  // step t+1
                    issue2(s0, t + 3);
                } else {
                    if (t + 1 < KSTEPS) commit(1, s1);
                    issue2(s1, t + 3);
                }
                lds_barrier();
            }
        }
    } else {
        // runtime K: one-step prefetch
        const int nt = K / KROW;
        Slot s0;
        issue(s0, 0);
        commit(0, s0);
        lds_barrier();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) issue(s0, t + 1);
            compute(buf);
            if (t + 1 < nt) commit(buf ^ 1, s0);
            lds_barrier();
        }
    }
}

// ---- LDS-DMA main loop -----------------------------------------------------------------------------------------
// Same tile image and MFMA step as gemm_mainloop, but the operand tiles go global -> LDS by `buffer_load ... lds`
// (16 B per lane, 1 KiB = 8 LDS rows per wave instruction): no register slot, no ds_write, one address per KiB.
// The DMA writes LDS linearly (wave base + lane * 16), so the chunk swizzle of lds_off() is applied on the SOURCE
// side: lane L of the instruction that fills rows R .. R+7 fetches chunk (L & 7) ^ sw(row) of row R + (L >> 3).
// NSTAGE LDS buffers form a ring with NSTAGE-1 K steps in flight; per step one counted `s_waitcnt vmcnt(N)` (own
// loads of the step about to be read) and ONE raw s_barrier (everybody's loads landed + everybody is done reading
// the buffer that is refilled next). Buffer descriptors are based at the tile's first row with the remaining
// bytes of the matrix as the bound, so rows past M / N read as zeros and per-lane offsets stay 32-bit.
#define OCM_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Workgroup barrier that leaves the vector-memory counter alone (LDS-DMA stays in flight across it) but retires
// this wave's own LDS reads first: once every wave has passed it, nobody is still reading the stage that the next
// DMA overwrites (WAR), and — after the counted vmcnt wait in front of it — everybody's DMA of the stage read next
// has landed (RAW).
__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// s_waitcnt vmcnt(younger * LPS) for 0 <= younger <= S (the count must be an immediate)
template <int LPS, int S>
__device__ __forceinline__ void wait_vm_steps(int younger) {
    if constexpr (S == 0) {
        OCM_VMCNT(0);
    } else {
        if (younger >= S)
            OCM_VMCNT(S * LPS);
        else
            wait_vm_steps<LPS, S - 1>(younger);
    }
}

// CNT buffer_load ... lds instructions of one wave: instruction j fills LDS rows (j * NW + wave) * 8 .. + 7 of the image
// at `img` from the per-lane byte offsets voff[j] (+ soff, the K step) of the buffer `rs`.
template <int CNT, int NW, int AUX = 0>
__device__ __forceinline__ void dma_rows(__amdgpu_buffer_rsrc_t rs, char *img, const int (&voff)[CNT], int wave,
                                         int soff) {
#if defined(__HIP_DEVICE_COMPILE__)  // hipcc's HOST pass mis-handles a second instantiation context of this builtin
    typedef __attribute__((address_space(3))) void *lds_ptr;
#pragma unroll
    for (int j = 0; j < CNT; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(img + (j * NW + wave) * 1024), 16, voff[j], soff, 0, AUX);
#endif
}

// the same with an explicit LDS piece index per instruction
template <int CNT, int AUX = 0>
__device__ __forceinline__ void dma_pieces(__amdgpu_buffer_rsrc_t rs, char *img, const int (&voff)[CNT], const int (&piece)[CNT],
                                           int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
#pragma unroll
    for (int j = 0; j < CNT; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(img + piece[j] * 1024), 16, voff[j], soff, 0, AUX);
#endif
}

// A_AUX: cache-policy bits of the A operand's DMA loads (2 = nt: streamed activations should not evict the weights)
template <class Cfg, class E, bool SWAP, int KSTEPS, int NSTAGE, int A_AUX = 0>
__device__ __forceinline__ void gemm_mainloop_dma(const E *__restrict__ A, int64_t lda, const E *__restrict__ W,
                                                  int64_t ldw, int m0, int n0, int M, int N, int K, char *smem,
                                                  f32x16 (&acc)[Cfg::TM][Cfg::TN], const float *__restrict__ bias) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, NW = NT / 64;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    // DMA instructions per wave per K step. A tile whose row count is not a multiple of 8 NW (HALF: 160 rows = 20 pieces on eight
    // waves) gives every wave ceil(pieces / NW) instructions: the surplus ones repeat the piece of wave - (pieces % NW) — the same
    // bytes to the same LDS rows twice, a benign duplicate that keeps the counted vmcnt waits the same in every wave
    constexpr int A_P = BM / 8, A_I = (A_P + NW - 1) / NW, B_I = BN / 8 / NW, LPS = A_I + B_I;
    constexpr int STAGE = (BM + BN) * 128, D = NSTAGE - 1;
    static_assert(BM % 8 == 0 && BN % (8 * NW) == 0 && (A_P % NW == 0 || 2 * (A_P % NW) >= NW), "each wave fills whole groups of 8 rows");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 LDS stages");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int r = lane & 31, h = lane >> 5;
    const int esz = (int)sizeof(E);
    const int nsteps = KSTEPS > 0 ? KSTEPS : K / Elem<E>::KROW;

    auto bound = [](int64_t rows, int64_t ld_bytes) {
        const int64_t b = rows * ld_bytes;
        return (unsigned)(b > 0xFFFFFFFFll ? 0xFFFFFFFFll : b);
    };
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)A + (int64_t)m0 * lda * esz), 0,
                                                       bound(M - m0, lda * esz), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)W + (int64_t)n0 * ldw * esz), 0,
                                                       bound(N - n0, ldw * esz), 0x00020000);
    int voffA[A_I], voffB[B_I];
    int pieceA[A_I];  // LDS piece (8 rows) that instruction j of this wave fills
#pragma unroll
    for (int j = 0; j < A_I; ++j) {
        int pc = j * NW + wave;
        if (pc >= A_P) pc -= A_P % NW;  // surplus instruction of the last round: a duplicate (see above)
        pieceA[j] = pc;
        const int rho = pc * 8 + (lane >> 3);
        voffA[j] = rho * (int)(lda * esz) + (((lane & 7) ^ ((rho >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < B_I; ++j) {
        const int rho = (j * NW + wave) * 8 + (lane >> 3);
        voffB[j] = rho * (int)(ldw * esz) + (((lane & 7) ^ ((rho >> 1) & 7)) << 4);
    }
    // K step t -> stage buf (a macro, not a lambda)
#define OCM_DMA_ISSUE(t, buf)                                                   \
    do {                                                                        \
        char *st_ = smem + (buf) * STAGE;                                       \
        dma_pieces<A_I, A_AUX>(rsA, st_, voffA, pieceA, (t) * 128);             \
        dma_rows<B_I, NW>(rsB, st_ + BM * 128, voffB, wave, (t) * 128);         \
    } while (0)
    auto compute = [&](int buf) {
        const char *st = smem + buf * STAGE;
        mma_step<Cfg, E, SWAP>(st + (wm * 32) * 128, st + BM * 128 + (wn * 32) * 128, r, h, acc, wm);
    };

    // accumulators start at the bias (see gemm_mainloop): the bias loads are issued first (oldest), the prologue
    // DMAs next, and the accumulators are filled while those are in flight
    constexpr int ACCL = accl_of<Cfg, E>();
    constexpr int NBV = SWAP ? 16 : (ACCL ? 2 : 1);  // distinct bias values a lane needs per tile
    float bv[TN][NBV];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + (j * Cfg::WAVES_N + wn) * 32;
#pragma unroll
        for (int e = 0; e < NBV; ++e)
            bv[j][e] = bias ? bias[min(nb + (SWAP ? acc_rpos<ACCL>(e, lane) : acc_cpos<ACCL>(4 * e, lane)), N - 1)] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (t < nsteps) OCM_DMA_ISSUE(t, t);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = bv[j][SWAP ? e : (ACCL ? ((e >> 2) & 1) : 0)];
    int bc = 0, bi = D % NSTAGE;  // stage computed next / stage filled next
#ifdef OCM_GEMM_STAMPS
    unsigned long long stamp_wait = 0;
#endif
    for (int t = 0; t < nsteps; ++t) {
        // own DMAs of step t have landed once at most the (younger) steps t+1 .. t+D-1 are pending
        const int younger = min(D - 1, nsteps - 1 - t);
#ifdef OCM_GEMM_STAMPS
        const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
        WSTAMP(0);  // step entered
        wait_vm_steps<LPS, NSTAGE - 2>(younger);
        WSTAMP(1);  // own DMAs landed
#ifdef OCM_GEMM_STAMPS
        const unsigned long long tw1 = __builtin_readcyclecounter();
#endif
        raw_barrier();
#ifdef OCM_GEMM_STAMPS
        if (t == 0) STAMP(6);  // first tile landed
        if (t > 0) {  // slot 7: cycles this wave spent waiting for its own DMAs (low 32 bits) and at the barrier (high 32 bits), steps 1..
            const unsigned long long tw2 = __builtin_readcyclecounter();
            stamp_wait += ((tw2 - tw1) << 32) | ((tw1 - tw0) & 0xFFFFFFFFull);
        }
#endif
        WSTAMP(2);  // barrier passed
#if !defined(OCM_ABL) || OCM_ABL != 2  // ablation 2: no DMA after the prologue
        if (t + D < nsteps) OCM_DMA_ISSUE(t + D, bi);
#endif
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(3);  // DMA issued
        compute(bc);
#ifdef OCM_GEMM_STAMPS
        if (t == nsteps / 2) {  // wait for the last MFMA's result, then stamp
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("v_mov_b32 %0, %0\n\ts_nop 0" : "+v"(acc[TM - 1][TN - 1][15]));
            WSTAMP(4);
        }
#endif
        bc = bc + 1 == NSTAGE ? 0 : bc + 1;
        bi = bi + 1 == NSTAGE ? 0 : bi + 1;
    }
    raw_barrier();  // every wave is done reading: the epilogue may reuse the LDS
#ifdef OCM_GEMM_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + 7] = stamp_wait;
#endif
#undef OCM_DMA_ISSUE
}

// Per-row and per-column scalars of an epilogue (Epi::ROWTAB: the folded LayerNorm's (mu, rstd) per row and (c, d) per
// column). They are REQUESTED at kernel start — thread t < BM the row sums of row m0 + t (raw slots: the arithmetic on
// them waits until the epilogue, so that no wait for these loads sits in front of the main loop), thread t < BN the column
// pair of column n0 + t — and published in two small LDS tables behind the operand area, ahead of the barrier that also
// publishes the accumulator image.
constexpr int EPI_MAXS = 6;
struct EpiPre {
    f32x2 raw[EPI_MAXS];
    f32x2 cd;
};
struct EpiNoPre {};  // epilogues without tables carry nothing across the main loop
template <class Cfg, class Epi>
__device__ __forceinline__ auto epi_prefetch(const Epi &epi, int m0, int n0, int M, int N) {
    if constexpr (Epi::ROWTAB) {
        EpiPre p;
#pragma unroll
        for (int i = 0; i < EPI_MAXS; ++i) p.raw[i] = f32x2{0.f, 0.f};
        p.cd = f32x2{0.f, 0.f};
        if (threadIdx.x < Cfg::BM) epi.row_raw(min(m0 + (int)threadIdx.x, M - 1), p.raw);
        if (threadIdx.x < Cfg::BN) p.cd = epi.col_entry(min(n0 + (int)threadIdx.x, N - 1));
        return p;
    } else {
        return EpiNoPre();
    }
}

// LDS_AVAIL: bytes of LDS the kernel owns (the register-staged kernels: Cfg::LDS_BYTES; the LDS-DMA kernels: their ring)
template <class Cfg, bool SWAP, class Epi, int LDS_AVAIL = Cfg::LDS_BYTES, class Pre = EpiNoPre, int ACCL = 0>
__device__ __forceinline__ void run_epilogue(const f32x16 (&acc)[Cfg::TM][Cfg::TN], char *smem, const Epi &epi, int m0,
                                             int n0, bool active = true, const Pre *pre = nullptr) {
    f32x2 *rowtab = (f32x2 *)(smem + LDS_AVAIL), *coltab = rowtab + Cfg::BM;
    if constexpr (Epi::ROWTAB) {
        if (threadIdx.x < Cfg::BM) rowtab[threadIdx.x] = epi.row_final(pre->raw);
        if (threadIdx.x < Cfg::BN) coltab[threadIdx.x] = pre->cd;
    }
    // `active` false (workgroup-uniform per wave): a wave that holds no accumulators (the loader waves of the
    // wave-specialised kernel) only keeps the barrier count of the passes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int r = lane & 31, h = lane >> 5;
    if constexpr (Cfg::BM * Cfg::BN * 4 <= LDS_AVAIL) {
        constexpr int COLS = SWAP ? Cfg::BM : Cfg::BN;
        float *C = (float *)smem;
        if (active)
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                const bool half = Cfg::HALF && i == Cfg::TM - 1;  // the wave's half tile: 16 rows at wm * 16 of the bottom band
                const int tm = half ? (Cfg::TM - 1) * Cfg::RB + wm * 16 : (i * Cfg::WAVES_M + wm) * 32, tn = (j * Cfg::WAVES_N + wn) * 32;
                // normal: lane -> column n, registers -> rows m.  swapped: lane -> m, registers -> n.
                const int row0 = SWAP ? tn : tm, col0 = SWAP ? tm : tn;
#pragma unroll
                for (int e = 0; e < (half ? 8 : 16); ++e)
                    C[(row0 + acc_rpos<ACCL>(e, lane)) * COLS + col0 + acc_cpos<ACCL>(e, lane)] = acc[i][j][e];
            }
        STAMP(2);
        lds_barrier();
        STAMP(3);
        if (active) epi.template run<PassCfg<Cfg::BM, Cfg::BN, Cfg::NT>>((const float *)C, m0, n0, rowtab, coltab);
    } else if constexpr (!SWAP) {
        constexpr int PASS = Cfg::RB * Cfg::BN * 4;  // one band of RB rows
        static_assert(PASS <= LDS_AVAIL, "a row band must fit the operand LDS");
        constexpr int NBUF = 2 * PASS <= LDS_AVAIL ? 2 : 1;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
            if (NBUF == 1 && i) lds_barrier();  // the band's readers are done
            float *C = (float *)(smem + (i % NBUF) * PASS);
            const bool half = Cfg::HALF && i == Cfg::TM - 1;  // the bottom band of 16-row half tiles: RB / 2 rows
            if (active)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                const int col0 = (j * Cfg::WAVES_N + wn) * 32;
#pragma unroll
                for (int e = 0; e < (half ? 8 : 16); ++e)
                    C[(wm * (half ? 16 : 32) + acc_rpos<ACCL>(e, lane)) * Cfg::BN + col0 + acc_cpos<ACCL>(e, lane)] = acc[i][j][e];
            }
            lds_barrier();  // also orders pass i-2's reads of this buffer before pass i's writes (see above)
            if (active) {
                if (half)
                    epi.template run<PassCfg<Cfg::RB / 2, Cfg::BN, Cfg::NT>>((const float *)C, m0 + i * Cfg::RB, n0, rowtab + i * Cfg::RB, coltab);
                else
                    epi.template run<PassCfg<Cfg::RB, Cfg::BN, Cfg::NT>>((const float *)C, m0 + i * Cfg::RB, n0, rowtab + i * Cfg::RB, coltab);
            }
        }
    } else {
        static_assert(!Cfg::HALF, "half tiles: normal orientation only");
        constexpr int PASS = Cfg::CB * Cfg::BM * 4;  // one band of CB feature rows of the transposed tile
        static_assert(PASS <= LDS_AVAIL, "a column band must fit the operand LDS");
        constexpr int NBUF = 2 * PASS <= LDS_AVAIL ? 2 : 1;
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            if (NBUF == 1 && j) lds_barrier();
            float *C = (float *)(smem + (j % NBUF) * PASS);
            if (active)
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
                const int col0 = (i * Cfg::WAVES_M + wm) * 32;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    C[(wn * 32 + acc_rpos<ACCL>(e, lane)) * Cfg::BM + col0 + acc_cpos<ACCL>(e, lane)] = acc[i][j][e];
            }
            lds_barrier();
            if (active) epi.template run<PassCfg<Cfg::BM, Cfg::CB, Cfg::NT>>((const float *)C, m0, n0 + j * Cfg::CB, rowtab, coltab + j * Cfg::CB);
        }
    }
}

// LDS-DMA variant of gemm_kernel for row-major A (activations): dynamic LDS = NSTAGE * (BM + BN) * 128 bytes.
template <class Cfg, class E, bool SWAP, int KSTEPS, int NSTAGE, class Epi, int A_AUX = 0>
__global__ __launch_bounds__(Cfg::NT) void gemm_dma_kernel(const E *__restrict__ A, int64_t lda, const E *__restrict__ W,
                                                           int64_t ldw, int M, int N, int K, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    f32x16 acc[Cfg::TM][Cfg::TN];
    const auto pre = epi_prefetch<Cfg>(epi, m0, n0, M, N);
    STAMP(0);
#if defined(OCM_ABL) && OCM_ABL == 4  // ablation 4: epilogue only
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = (float)(threadIdx.x + e);
#else
    gemm_mainloop_dma<Cfg, E, SWAP, KSTEPS, NSTAGE, A_AUX>(A, lda, W, ldw, m0, n0, M, N, K, smem, acc, epi.bias);
#endif
    STAMP(1);
#if defined(OCM_ABL) && OCM_ABL == 3  // ablation 3: no epilogue (accumulators kept live)
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
    run_epilogue<Cfg, SWAP, Epi, NSTAGE * (Cfg::BM + Cfg::BN) * 128, decltype(pre), accl_of<Cfg, E>()>(acc, smem, epi, m0, n0, true, &pre);
#endif
    STAMP(4);
#ifdef OCM_GEMM_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(5);
#endif
}

// Generic kernel: grid = tiles_m * tiles_n workgroups (linear, XCD-remapped so the
// workgroups that share an A row panel sit on one XCD's L2).
template <class Cfg, class E, bool SWAP, int KSTEPS, class ALoad, class Epi>
__global__ __launch_bounds__(Cfg::NT) void gemm_kernel(ALoad al, const E *__restrict__ W, int64_t ldw, int M, int N,
                                                       int K, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / tiles_n, tn = id - tm * tiles_n;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    f32x16 acc[Cfg::TM][Cfg::TN];
    const auto pre = epi_prefetch<Cfg>(epi, m0, n0, M, N);
    STAMP(0);
    gemm_mainloop<Cfg, E, SWAP, KSTEPS>(al, W, ldw, m0, n0, M, N, K, smem, acc, epi.bias);
    STAMP(1);  // prologue + K loop done
    run_epilogue<Cfg, SWAP, Epi, Cfg::LDS_BYTES, decltype(pre), accl_of<Cfg, E>()>(acc, smem, epi, m0, n0, true, &pre);  // STAMP 2: accumulators staged, 3: barrier passed
    STAMP(4);  // epilogue body issued
#ifdef OCM_GEMM_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(5);  // stores acknowledged
#endif
}
