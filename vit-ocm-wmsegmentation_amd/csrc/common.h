// common.h — shared types and device helpers for the gfx950 (CDNA4) kernels.
// Wavefront = 64 lanes everywhere; MFMA operands are 8 x bf16 per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OCM_WAVE 64
#define OCM_HEAD_DIM 64

// Row of a 32x32 MFMA accumulator held in register `reg` by lane half `h`
// (C/D map of v_mfma_f32_32x32x16_bf16: col = lane & 31).
__device__ __forceinline__ int acc_row32(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// LDS tile image used by every MFMA operand tile here: rows of 64 bf16 (128 B),
// eight 16-B chunks per row, chunk index XOR-swizzled with (row >> 1) & 7 so that
// a ds_read_b128 of 32 rows x 2 chunks (one 32x32x16 operand) is bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Workgroup barrier that orders LDS traffic only: `s_waitcnt lgkmcnt(0); s_barrier`. Unlike
// __syncthreads() it does not drain the vector-memory counter, so global prefetches and epilogue
// stores stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// exact fp32 MFMA (OCM_PREC_FP32): one f32 per lane per operand, k = lane >> 5
__device__ __forceinline__ f32x16 mfma32f(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 cvt8(f32x4 lo, f32x4 hi) {
    bf16x8 r;
    r[0] = (bf16)lo[0]; r[1] = (bf16)lo[1]; r[2] = (bf16)lo[2]; r[3] = (bf16)lo[3];
    r[4] = (bf16)hi[0]; r[5] = (bf16)hi[1]; r[6] = (bf16)hi[2]; r[7] = (bf16)hi[3];
    return r;
}

// ---- split-bf16 pairs (OCM_PREC_BF16X3) -------------------------------------------------------------
// A value x is carried as two bf16 numbers hi = bf16(x), lo = bf16(x - hi): x = hi + lo up to 2^-17 |x|, and a
// product is evaluated as hi*hi' + hi*lo' + lo*hi' on the bf16 MFMA (three instructions, fp32 accumulate; the
// dropped lo*lo' term is 2^-18 relative). Memory / LDS layout of such a tensor: the contraction axis is cut into
// groups of 32 elements and each group is one 128-byte row segment [32 x hi | 32 x lo] — the same bytes per
// element as fp32, every 16-byte chunk is a ready MFMA fragment, and a 128-byte LDS row is one K step of 32.
struct sp32 {  // element tag: sizeof == 4 bytes per logical element
    uint32_t bits;
};
// byte offset of the hi half of element `col` inside its row (lo half: + 64)
__device__ __host__ __forceinline__ int sp_off(int col) { return (col >> 5) * 128 + (col & 31) * 2; }
__device__ __forceinline__ void split1(float x, bf16 &hi, bf16 &lo) {
    hi = (bf16)x;
    lo = (bf16)(x - (float)hi);
}
__device__ __forceinline__ void split4(const f32x4 &x, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16 t = (bf16)x[e];
        hi[e] = t;
        lo[e] = (bf16)(x[e] - (float)t);
    }
}
__device__ __forceinline__ void split8(const f32x4 &a, const f32x4 &b, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16 t = (bf16)a[e], u = (bf16)b[e];
        hi[e] = t;
        lo[e] = (bf16)(a[e] - (float)t);
        hi[4 + e] = u;
        lo[4 + e] = (bf16)(b[e] - (float)u);
    }
}
// acc += a*b for split operands: small terms first
__device__ __forceinline__ f32x16 mfma32x3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
#if defined(OCM_ABL) && OCM_ABL == 1  // ablation 1: operands fetched, no matrix instructions
    asm volatile("" ::"v"(ah), "v"(al), "v"(bh), "v"(bl));
    return c;
#endif
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

// The same product on v_mfma_f32_16x16x32_bf16 (gemm_core.h: GemmCfg::MF16): a 16 x 16 tile per instruction, K = 32 — one whole
// 128-byte LDS row segment of hi (or lo) halves per step. Lane l holds A[row l & 15][k = 8 (l >> 4) + j] and
// B[k = 8 (l >> 4) + j][col l & 15]; C/D: col = l & 15, row = 4 (l >> 4) + reg.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma16x3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
// Accumulator layouts of a 32 x 32 output tile held as f32x16 per lane. ACCL 0: one 32x32x16 accumulator (register -> row
// acc_row32, lane & 31 -> column). ACCL 1: four 16x16x32 accumulators, sub-tile q = 2 * (row half) + (column half) in
// registers 4q .. 4q+3. rpos = coordinate along the register-indexed axis, cpos = along the lane-indexed axis.
template <int ACCL>
__device__ __forceinline__ int acc_rpos(int e, int lane) {
    return ACCL ? 16 * (e >> 3) + 4 * ((lane >> 4) & 3) + (e & 3) : acc_row32(e, lane >> 5);
}
template <int ACCL>
__device__ __forceinline__ int acc_cpos(int e, int lane) {
    return ACCL ? 16 * ((e >> 2) & 1) + (lane & 15) : (lane & 31);
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive logical ids
// (which share an A row panel) land on one XCD / one L2. Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// The same remap for a 2-D grid (x fastest, as the dispatcher walks it): returns the logical (x, y).
// Every kernel of the layer chain maps token rows / (image, head) pairs to XCDs the same way — contiguous
// eighths — so a stage reads what the previous stage left in ITS XCD's L2 instead of going to the
// Infinity Cache for it.
__device__ __forceinline__ void xcd_remap2(int &bx, int &by) {
    const int id = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    by = id / (int)gridDim.x;
    bx = id - by * (int)gridDim.x;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// key held by accumulator register `reg` of lane half `h` when K rows are loaded in pi order
__device__ __forceinline__ int key_of_reg(int reg, int h) {
    return (reg & 3) + 4 * ((reg >> 2) & 1) + 8 * h + 16 * (reg >> 3);
}
__device__ __forceinline__ int pi_row(int r) {  // swap bits 2 and 3
    return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. fp32-grade) on the hardware
// reciprocal / exp2 units: ~12 VALU ops instead of libm erff's ~50, which made the fc1 epilogue
// cost more than its MFMA main loop.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}
// Two GELUs at once on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32): with erf(z) = 1 - P(t) e^{-z^2} (z >= 0)
//   gelu(x) = max(x, 0) - 0.5 |x| P(t) exp(-x^2 / 2),   t = 1 / (1 + p |x| / sqrt(2))
// (same A&S 7.1.26 coefficients as erf_as, the 0.5 folded in), ~10 VALU ops per element.
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 den = ax * 0.2316418882f + 1.0f;  // p / sqrt(2)
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 p = t * 0.5307027145f + (-0.7265760135f);  // a5 / 2, a4 / 2
    p = p * t + 0.7107068705f;                       // a3 / 2
    p = p * t + (-0.142248368f);                     // a2 / 2
    p = p * t + 0.127414796f;                        // a1 / 2
    p = p * t;
    const f32x2 u = x * 0.8493218003f;  // sqrt(log2(e) / 2)
    const f32x2 nu2 = -(u * u);
    const f32x2 e = {__builtin_amdgcn_exp2f(nu2[0]), __builtin_amdgcn_exp2f(nu2[1])};
    const f32x2 g = ax * p * e;
    const f32x2 r = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return r - g;
}

// GELU(approximate='none') of the reference (nn.GELU, dino/vision_transformer.py:53)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
