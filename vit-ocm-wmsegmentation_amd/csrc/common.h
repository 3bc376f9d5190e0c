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

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 cvt8(f32x4 lo, f32x4 hi) {
    bf16x8 r;
    r[0] = (bf16)lo[0]; r[1] = (bf16)lo[1]; r[2] = (bf16)lo[2]; r[3] = (bf16)lo[3];
    r[4] = (bf16)hi[0]; r[5] = (bf16)hi[1]; r[6] = (bf16)hi[2]; r[7] = (bf16)hi[3];
    return r;
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive logical ids
// (which share an A row panel) land on one XCD / one L2. Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
