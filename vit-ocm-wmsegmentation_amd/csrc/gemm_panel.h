// gemm_panel.h — "row panel" bf16 MFMA GEMM for the short-K projections (K = D <= 384).
//
//   C[m][n] = sum_k A[m][k] * W[n][k],   A = LayerNorm output [M][K] bf16,  W = nn.Linear weight [N][K]
//
// Why: with K = 384 a classic LDS-tiled GEMM re-reads the A tile for every column tile and needs
// (BM+BN)*128 B per 64-deep K step; at 128 x 128 that is 64 B/clk/CU from L2 for peak MFMA rate,
// twice what the L2 -> CU path sustains (measured: the tiled main loop ran at 7-11 TB/s of L2
// traffic, ~20 % of the MFMA peak). Here a workgroup of NW waves owns 32*NW rows; every wave keeps
// the MFMA A-operand fragments of its 32 rows for the WHOLE K extent in registers (K/16 x 4 VGPRs =
// 96 VGPRs at K = 384, loaded once), and the workgroup walks a range of 64-column weight tiles:
// per K step only an 8 KiB W tile crosses L2 -> LDS and is shared by all waves. Accumulators
// (32 x 64 per wave) are emitted per column tile through a wave-private LDS staging slab with 16-B stores.
//
// Pipeline per K step t (one LDS-only barrier):
//     read the B fragments of tile t+1 (LDS -> second fragment register set)
//     8 MFMAs on the fragments of tile t (already in registers: no LDS wait in front of the MFMAs)
//     commit tile t+2 from its staging registers to LDS buffer t & 1, refill them with tile t+5
// W tiles are therefore requested 3 steps before they are committed (L2 latency under load is
// ~1 us, i.e. several steps) and every load / LDS write is unconditional so that hipcc can count its
// own `s_waitcnt vmcnt(N)` (a conditional stream degrades every wait to vmcnt(0)).
//
// Grid = row_blocks x nsplit; block (rb, sp) computes rows [BM rb, BM rb + BM) x the sp-th share of
// the N/64 column tiles (balanced contiguous partition).
#pragma once
#include <type_traits>

#include "common.h"

#define PANEL_BN 64
#define PANEL_CPAD 68  // floats per staged C row (64 + 4: de-phases the 256-B bank window)

template <int NW>
struct PanelCfg {
    static constexpr int BM = 32 * NW, NT = 64 * NW;
    static constexpr int WCH = 512 / NT;  // 16-B chunks of a W tile per thread (1 at 8 waves, 2 at 4)
    // LDS: W ring 2 x 8 KiB, then NW wave-private C slabs of 32 x PANEL_CPAD floats
    static constexpr int LDS_BYTES = 2 * 8192 + NW * 32 * PANEL_CPAD * 4;
};

#ifdef PANEL_STAMP
// diagnostic build only: s_memtime stamps of workgroup 0 / wave 0 (never in the shipped library)
__device__ unsigned long long g_panel_stamps[512];
#define STAMP(idx)                                                                          \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (idx) < 512) g_panel_stamps[(idx)] = t_; \
    } while (0)
#else
#define STAMP(idx) \
    do {           \
    } while (0)
#endif

// Epilogue concept (mw = first row of the calling wave, n0 = first column of the 64-wide tile):
//   bool swap_for(int n0) const      whether the tile at column n0 wants a transposed accumulator
//   void init(acc, swapped, n0, lane) accumulator start value (the bias: saves the add per element)
//   void stage(acc, swapped, slab, lane)   accumulators -> wave-private LDS slab
//   quarter p = 0..3 of the slab -> activation / cast / store, in three pieces so that the kernel can
//   interleave it with MFMAs:  part_begin(state, p, swapped, slab, lane)   slab -> registers
//                              part_elem(state, j), j = 0..7                per-element math (GELU)
//                              part_end(state, p, valid, swapped, mw, n0, lane)   cast + predicated stores
// With K = 384 an output element receives only 384 MACs, so the per-element epilogue work (GELU is
// ~70 issue cycles per element and lane) rivals the MFMA time. The epilogue of column tile j is
// therefore software-pipelined: its accumulators are parked in the slab right after the last MFMA and
// the four quarters are processed inside K steps 0..3 of tile j+1, where the VALU is otherwise idle.
template <int KD, int NW, class Epi>  // K = 64 * KD
__global__ __launch_bounds__(64 * NW, 2) void panel_gemm_kernel(const bf16 *__restrict__ A, const bf16 *__restrict__ W,
                                                                int M, int N, int nsplit, Epi epi) {
    typedef PanelCfg<NW> Cfg;
    static_assert(KD % 2 == 0 && KD >= 4, "panel GEMM needs an even number (>= 4) of 64-wide K steps");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 64 * KD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x / nsplit, sp = blockIdx.x - rb * nsplit;
    const int ntiles = N / PANEL_BN;
    const int jt0 = (int)((long)ntiles * sp / nsplit), jt1 = (int)((long)ntiles * (sp + 1) / nsplit);
    const int mw = rb * Cfg::BM + wave * 32;  // first row of this wave
    char *Ws = smem;
    float *slab = (float *)(smem + 2 * 8192) + wave * 32 * PANEL_CPAD;
    const int nsteps = (jt1 - jt0) * KD;
    if (nsteps <= 0) return;  // workgroup-uniform (more splits than column tiles)
    STAMP(0);

    // ---- W staging: tile index i -> column tile jt0 + i / KD, K step i % KD. Indices past the end re-read
    // the last tile (harmless duplicates into a free buffer) to keep the stream unconditional.
    struct WRegs {
        bf16x8 c[Cfg::WCH];
    };
    int woff[Cfg::WCH];
    const bf16 *wsrc[Cfg::WCH];
#pragma unroll
    for (int i = 0; i < Cfg::WCH; ++i) {
        const int q = tid + Cfg::NT * i, row = q >> 3, c = q & 7;
        woff[i] = lds_off(row, c);
        wsrc[i] = W + (int64_t)row * K + c * 8;
    }
    auto wload = [&](int i) -> WRegs {
        i = min(i, nsteps - 1);
        const int jt = jt0 + i / KD, ks = i - (i / KD) * KD;
        const int64_t off = (int64_t)jt * PANEL_BN * K + ks * 64;
        WRegs w;
#pragma unroll
        for (int j = 0; j < Cfg::WCH; ++j) w.c[j] = *(const bf16x8 *)(wsrc[j] + off);
        return w;
    };
    auto wcommit = [&](int buf, const WRegs &w) {
#pragma unroll
        for (int j = 0; j < Cfg::WCH; ++j) *(bf16x8 *)(Ws + buf * 8192 + woff[j]) = w.c[j];
    };
    WRegs w0 = wload(0), w1 = wload(1);

    // ---- A fragments, whole K, straight to registers (rows clamped: out-of-range rows are never stored)
    bf16x8 af[KD * 4];
    {
        const bf16 *ap = A + (int64_t)min(mw + r, M - 1) * K + 8 * h;
#pragma unroll
        for (int s = 0; s < KD * 4; ++s) af[s] = *(const bf16x8 *)(ap + 16 * s);
    }

    struct BFrag {
        bf16x8 f[4][2];
    };
    auto read_frags = [&](int buf, BFrag &bf) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
                bf.f[s][jn] = *(const bf16x8 *)(Ws + buf * 8192 + jn * 32 * 128 + lds_off(r, 2 * s + h));
    };

    // prologue: tiles 0 and 1 into LDS, fragments of tile 0 into registers, tiles 2..4 in flight
    BFrag fa, fb;
    wcommit(0, w0);
    wcommit(1, w1);
    WRegs s0 = wload(2), s1 = wload(3), s2 = wload(4);  // consumed in the order s0, s1, s2, s0, ...
    lds_barrier();
    read_frags(0, fa);
    lds_barrier();
    STAMP(1);
    int stamp_i = 2;
    (void)stamp_i;

    f32x16 acc[2];
    bool pending = false, pswapped = false;  // epilogue of the previous column tile still to be emitted
    int pn0 = 0;
    auto step = [&](int t, int ks, BFrag &cur, BFrag &nxt, WRegs &slot, auto swp) {
        constexpr bool swapped = decltype(swp)::value;
        const bool epi_step = (ks % KD) < 4;  // compile-time after unrolling
        read_frags((ks + 1) & 1, nxt);
        // Quarter (ks) of the previous tile's epilogue rides in this step: its 8 per-lane elements are
        // evaluated one per MFMA, in program order (sched_barrier pins it), so the activation math issues
        // in the matrix pipe's shadow instead of after the MFMA cluster. No branch: `pending` only
        // predicates the stores.
        typename Epi::PartState ps;
        if (epi_step) epi.part_begin(ps, ks % KD, pswapped, slab, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 a = af[ks * 4 + s];  // ks is a compile-time constant after unrolling
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                acc[jn] = swapped ? mfma32(cur.f[s][jn], a, acc[jn]) : mfma32(a, cur.f[s][jn], acc[jn]);
                if (epi_step && Epi::kHasElem) {
                    epi.part_elem(ps, 2 * s + jn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        wcommit(ks & 1, slot);  // tile t + 2 -> buffer t & 1 (its previous tenant, tile t, is in `cur`)
        slot = wload(t + 5);
        if (epi_step) epi.part_end(ps, ks % KD, pending, pswapped, mw, pn0, lane);
        lds_barrier();
        STAMP(stamp_i);
        ++stamp_i;
    };
    // one column tile = KD steps. With KD % 6 == 0 the buffer / fragment-set parity (ks & 1) and the
    // staging slot (ks % 3) are compile-time; otherwise the three slots rotate through registers.
    auto ktile = [&](int base, auto swp) {
        if constexpr (KD % 6 == 0) {
#pragma unroll
            for (int ks = 0; ks < KD; ks += 6) {
                step(base + ks + 0, ks + 0, fa, fb, s0, swp);
                step(base + ks + 1, ks + 1, fb, fa, s1, swp);
                step(base + ks + 2, ks + 2, fa, fb, s2, swp);
                step(base + ks + 3, ks + 3, fb, fa, s0, swp);
                step(base + ks + 4, ks + 4, fa, fb, s1, swp);
                step(base + ks + 5, ks + 5, fb, fa, s2, swp);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KD; ks += 2) {
                step(base + ks, ks, fa, fb, s0, swp);
                step(base + ks + 1, ks + 1, fb, fa, s1, swp);
                const WRegs t0 = s0, t1 = s1;  // rotate: next two steps consume s2, then the refilled s0
                s0 = s2;
                s1 = t0;
                s2 = t1;
            }
        }
    };

    for (int jt = jt0; jt < jt1; ++jt) {
        const int n0 = jt * PANEL_BN;
        const bool swapped = epi.swap_for(n0);  // workgroup-uniform
        epi.init(acc, swapped, n0, lane);
        const int base = (jt - jt0) * KD;
        if (swapped)  // two copies of the K loop, so the MFMA operand order is static in each
            ktile(base, std::true_type{});
        else
            ktile(base, std::false_type{});
        if (mw < M) {  // wave-uniform
            epi.stage(acc, swapped, slab, lane);
            pending = true;
            pswapped = swapped;
            pn0 = n0;
        }
        STAMP(stamp_i);
        ++stamp_i;
    }
    if (pending) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            typename Epi::PartState ps;
            epi.part_begin(ps, p, pswapped, slab, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) epi.part_elem(ps, j);
            epi.part_end(ps, p, true, pswapped, mw, pn0, lane);
        }
    }
}
