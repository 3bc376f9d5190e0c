// kernels_attn.hip — multi-head self-attention with per-head attention-map extraction.
// Reference arithmetic: Attention.forward, dino/vision_transformer.py:83-87
//     attn = softmax((q @ k^T) * scale);  x = attn @ v
// and the consumers of its rows, utils.py:229-235 (compute_attention).
//
// Layouts (produced by the qkv GEMM epilogue, kernels_gemm.hip):
//     q, k : bf16 [B*H][n_pad][64]      (head_dim contiguous)
//     vt   : bf16 [B*H][64][n_pad]      (key index contiguous = V^T)
// head_dim is 64 for ViT-T/S/B.
//
// attn_fwd_kernel — flash-style: one wavefront owns 32 query rows, the workgroup's 4 waves
//   share 64-key K / V^T tiles staged through double-buffered swizzled LDS. Scores are computed
//   TRANSPOSED (S^T = K·Q^T, keys in registers, query on the lane) so that
//     * the row max / row sum are in-register reductions plus one lane<->lane+32 exchange, and
//     * the exponentiated tile is already the B operand of O^T += V^T·P^T (no LDS round trip):
//       K rows are fed in the order pi(r) (bits 2,3 of r swapped) which makes registers
//       8s..8s+7 of the accumulator hold keys 16s+8h..16s+8h+7, i.e. the MFMA k order.
//   Softmax is online (running max / sum, fp32, exp2 domain); P never touches HBM.
//   Optionally emits lse2[row] = max + log2(sum) so the probabilities can be re-materialised.
//
// attn_probs_kernel — the (B,H,N,N) fp32 probabilities the reference returns: recomputes
//   S = Q·K^T with the KEY on the lane, so each store instruction writes 32 consecutive
//   floats of one row; p = exp2(s*scale*log2e - lse2[row]).
//
// attn_rows_kernel — only selected query rows with the CLS column dropped (what
//   compute_attention consumes): one wavefront per (row, b, h), fp32 dot products.
#include <stdlib.h>

#include "launch.h"

#define LOG2E 1.4426950408889634f
int ocm_wt_mask();  // kernels_gemm.hip
#include "dev_knobs.h"
#define OCM_VMCNT_ATTN(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Deferred running maximum of the split-bf16 flash kernels (cdna_hip_programming.md T13). The reference point m of a query row
// only moves when some row of the wavefront found a score more than OCM_DEFER_MAX (log2 units) above its own m; until then the
// tile's exponentials are taken against the old m — values up to 2^OCM_DEFER_MAX, exact in fp32 and as split pairs (this mode keeps
// P at 2^-17 relative whatever its magnitude; a single-bf16 P would lose three bits) — and neither O nor l is rescaled. Any
// reference point gives the same softmax (lse2 = m + log2 l; the key-split merge and the probability kernels only see lse2 / (m, l)),
// so results change in the last bits only. With near-uniform attention the true maximum creeps up in most tiles of a long
// sequence (at N = 2305 some row of a wavefront moved in more than half of the 73 tiles): 16 packed multiplies of O and the
// branch around them per tile. Returns the factor for O and l (1 when nothing moved); m is updated in place.
#ifndef OCM_DEFER_MAX
#define OCM_DEFER_MAX 8.0f  // (0: the reference point follows every new maximum, the round-3 behaviour — A/B builds)
#endif
__device__ __forceinline__ float defer_max_update(float &m, float cand) {
    if (__any(cand > m + OCM_DEFER_MAX)) {  // m = -inf in front of the first tile: true
        const float mn = fmaxf(m, cand);
        const float alpha = fast_exp2(m - mn);
        m = mn;
        return alpha;
    }
    return 1.0f;
}

template <bool WANT_O, int HD = 64>
// compiled for three waves per SIMD (<= 168 registers, no spills): the softmax VALU work of one wave overlaps the
// MFMAs of the others (+17..24 % over the two-wave allocation hipcc picks by itself)
// (HD = 128, round 4 — the reference's build_model() encoder in single-bf16 precision: the K tile is two [64 keys][128 B] images,
// V^T has 128 rows, four context accumulators: two waves per SIMD)
__global__ __launch_bounds__(256, HD == 64 ? 3 : 2) void attn_fwd_kernel(const bf16 *__restrict__ Q, const bf16 *__restrict__ Kk,
                                                       const bf16 *__restrict__ Vt, bf16 *__restrict__ ctx,
                                                       float *__restrict__ lse2, int N, int npad, int H,
                                                       float scale2) {
    constexpr int TB = HD * 128, NS = HD / 16, NDB = HD / 32, CPT = HD / 32;  // tile bytes (K and V^T alike); chunks per thread
    static_assert(HD == 64 || HD == 128, "head widths built on MFMA");
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TB];  // K[2] | Vt[2], 8 KiB each at HD = 64
    char *Ks = smem, *Vs = smem + 2 * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * 4 + wave) * 32;
    const bool active = q0 < N;  // wave-uniform
    const bf16 *Qb = Q + (int64_t)bh * npad * HD;
    const bf16 *Kb = Kk + (int64_t)bh * npad * HD;
    const bf16 *Vb = Vt + (int64_t)bh * HD * npad;

    // Q^T as the B operand: lane (query r, half h) holds Q[q0+r][16s + 8h .. +7]
    bf16x8 qf[NS];
    {
        const int qrow = min(q0 + r, N - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8 *)(Qb + (int64_t)qrow * HD + 16 * s + 8 * h);
    }

    // staging: 8 HD K chunks + 8 HD V^T chunks of 16 B per tile, 256 threads -> HD / 32 each. K chunk qd: key qd / (HD / 8), its
    // 16-byte piece kc of the key's row -> image kc >> 3 ([64 keys][128 B] each), chunk kc & 7; V^T chunk qd: row d = qd >> 3
    bf16x8 rk[CPT], rv[CPT];
    auto issue = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int qd = tid + 256 * i, row = qd >> 3, c = qd & 7;
            const int krow = qd / (HD / 8), kc = qd % (HD / 8);
            const int key = min(kt * 64 + krow, N - 1);
            rk[i] = *(const bf16x8 *)(Kb + (int64_t)key * HD + kc * 8);
            const int key0 = kt * 64 + c * 8;  // V^T row = d, chunk = 8 keys
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (WANT_O && key0 < npad) {  // the statistics-only variant never touches V^T
                v = *(const bf16x8 *)(Vb + (int64_t)row * npad + key0);
                if (key0 + 8 > N) {  // keys >= N are padding: force exact zeros (0 * garbage must not be NaN)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key0 + e >= N) v[e] = (bf16)0.f;
                }
            }
            rv[i] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int qd = tid + 256 * i, row = qd >> 3, c = qd & 7;
            const int krow = qd / (HD / 8), kc = qd % (HD / 8);
            *(bf16x8 *)(Ks + buf * TB + (kc >> 3) * 8192 + lds_off(krow, kc & 7)) = rk[i];
            *(bf16x8 *)(Vs + buf * TB + lds_off(row, c)) = rv[i];
        }
    };

    f32x16 O[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int pr = pi_row(r);
    const int ntiles = (N + 63) >> 6;

    issue(0);
    commit(0);
    lds_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) issue(kt + 1);
        if (active) {
            const char *Kt = Ks + buf * TB, *Vtile = Vs + buf * TB;
            f32x16 S[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int e = 0; e < 16; ++e) S[sub][e] = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bf16x8 a = *(const bf16x8 *)(Kt + (s >> 2) * 8192 + sub * 32 * 128 + lds_off(pr, 2 * (s & 3) + h));
                    S[sub] = mfma32(a, qf[s], S[sub]);
                }
            }
            if ((kt + 1) * 64 > N) {  // tail tile: padding keys -> -inf (wave-uniform branch)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kt * 64 + sub * 32 + key_of_reg(e, h) >= N) S[sub][e] = -INFINITY;
            }
            // max on the raw scores (scale > 0 commutes with max), scale folded into the exp2 argument
            float mx = fmaxf(S[0][0], S[1][0]);
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(S[0][e], S[1][e]));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx * scale2);  // finite: every tile holds at least one valid key
            const float alpha = fast_exp2(m - mn);
            m = mn;
            float ps = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = fast_exp2(fmaf(S[sub][e], scale2, -mn));
                    S[sub][e] = p;
                    ps += p;
                }
            l = fmaf(l, alpha, ps);
            if (WANT_O) {
                if (__any(alpha != 1.0f)) {  // the running max moved somewhere in this wave
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int e = 0; e < 16; ++e) O[db][e] *= alpha;
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pb;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pb[e] = (bf16)S[sub][8 * s2 + e];
#pragma unroll
                        for (int db = 0; db < NDB; ++db) {
                            const bf16x8 a =
                                *(const bf16x8 *)(Vtile + db * 32 * 128 + lds_off(r, 4 * sub + 2 * s2 + h));
                            O[db] = mfma32(a, pb, O[db]);
                        }
                    }
            }
        }
        if (kt + 1 < ntiles) commit(buf ^ 1);
        lds_barrier();
    }

    if (!active) return;
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < N) {
        if (lse2 && h == 0) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
        if (WANT_O) {
            const float inv = 1.0f / lt;
            const int b = bh / H, head = bh - b * H;
            bf16 *dst = ctx + ((int64_t)b * N + qrow) * (H * HD) + head * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)(O[db][4 * g + e] * inv);
                    *(bf16x4 *)(dst + db * 32 + 8 * g + 4 * h) = o;
                }
        }
    }
}

// Whole-sequence variant for N <= 256 tokens (ViT-S/16 and ViT-B/16 at 224^2: N = 197): one workgroup
// of 8 waves per (batch, head); ALL keys / values of that head are brought into LDS by one burst of
// loads (K: 32 KiB as [256][64], V^T: 32 KiB as 4 blocks of [64 d][64 keys]) behind a single barrier,
// then every wave walks the key tiles of its 32 query rows with no further synchronisation. Compared
// with the streaming kernel this loads K/V once per head instead of once per 128 queries and removes
// the per-tile load -> LDS -> barrier latency chain (4 exposed round trips at N = 197).
template <bool WANT_O>
__global__ __launch_bounds__(512, 4) void attn_small_kernel(const bf16 *__restrict__ Q, const bf16 *__restrict__ Kk,
                                                            const bf16 *__restrict__ Vt, bf16 *__restrict__ ctx,
                                                            float *__restrict__ lse2, int N, int npad, int H,
                                                            float scale2) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 256 * 128];  // K | Vt, 32 KiB each
    char *Ks = smem, *Vs = smem + 256 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bh = xcd_remap(blockIdx.x, gridDim.x);
    const int q0 = wave * 32;
    const bf16 *Qb = Q + (int64_t)bh * npad * 64;
    const bf16 *Kb = Kk + (int64_t)bh * npad * 64;
    const bf16 *Vb = Vt + (int64_t)bh * 64 * npad;
    const int ntiles = (N + 63) >> 6;

    bf16x8 qf[4];
    {
        const int qrow = min(q0 + r, N - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8 *)(Qb + (int64_t)qrow * 64 + 16 * s + 8 * h);
    }
    {
        bf16x8 rk[4], rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = tid + 512 * i, row = qd >> 3, c = qd & 7;  // K row (key), 16-B chunk
            rk[i] = *(const bf16x8 *)(Kb + (int64_t)min(row, N - 1) * 64 + c * 8);
            // V^T: block kb = i (64 keys), row d = (tid >> 3), chunk = 8 keys
            const int d = tid >> 3, key0 = i * 64 + c * 8;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (WANT_O && key0 < npad) {  // the statistics-only variant never touches V^T
                v = *(const bf16x8 *)(Vb + (int64_t)d * npad + key0);
                if (key0 + 8 > N) {  // padding keys must be exact zeros: P = 0 times garbage may be NaN
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key0 + e >= N) v[e] = (bf16)0.f;
                }
            }
            rv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = tid + 512 * i, row = qd >> 3, c = qd & 7;
            *(bf16x8 *)(Ks + lds_off(row, c)) = rk[i];
            *(bf16x8 *)(Vs + i * 8192 + lds_off(tid >> 3, c)) = rv[i];
        }
    }
    lds_barrier();
    if (q0 >= N) return;  // no barrier below

    f32x16 O[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) O[0][e] = O[1][e] = 0.f;
    float m = -INFINITY, l = 0.f;  // m is in the scaled (log2) domain
    const int pr = pi_row(r);
    for (int kt = 0; kt < ntiles; ++kt) {
        const char *Kt = Ks + kt * 8192, *Vtile = Vs + kt * 8192;
        f32x16 S[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 a = *(const bf16x8 *)(Kt + sub * 32 * 128 + lds_off(pr, 2 * s + h));
                S[sub] = mfma32(a, qf[s], S[sub]);
            }
        }
        if ((kt + 1) * 64 > N) {  // tail tile: padding keys -> -inf (wave-uniform branch)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kt * 64 + sub * 32 + key_of_reg(e, h) >= N) S[sub][e] = -INFINITY;
        }
        // max on the raw scores (scale > 0 commutes with max), scale folded into the exp2 argument
        float mx = fmaxf(S[0][0], S[1][0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(S[0][e], S[1][e]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx * scale2);
        const float alpha = fast_exp2(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(fmaf(S[sub][e], scale2, -mn));
                S[sub][e] = p;
                ps += p;
            }
        l = fmaf(l, alpha, ps);
        if (WANT_O) {
            if (__any(alpha != 1.0f)) {  // the running max moved somewhere in this wave
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    O[0][e] *= alpha;
                    O[1][e] *= alpha;
                }
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pb[e] = (bf16)S[sub][8 * s2 + e];
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const bf16x8 a = *(const bf16x8 *)(Vtile + db * 32 * 128 + lds_off(r, 4 * sub + 2 * s2 + h));
                        O[db] = mfma32(a, pb, O[db]);
                    }
                }
        }
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < N) {
        if (lse2 && h == 0) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
        if (WANT_O) {
            const float inv = 1.0f / lt;
            const int b = bh / H, head = bh - b * H;
            bf16 *dst = ctx + ((int64_t)b * N + qrow) * (H * 64) + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)(O[db][4 * g + e] * inv);
                    *(bf16x4 *)(dst + db * 32 + 8 * g + 4 * h) = o;
                }
        }
    }
}

static hipError_t launch_attention_bf16(const bf16 *q, const bf16 *k, const bf16 *vt, bf16 *ctx, float *lse2, int batch,
                                        int n_tokens, int n_pad, int heads, float scale, hipStream_t s, int head_dim = 64) {
    if (head_dim == 128) {  // 128-wide heads: the streaming kernel at every sequence length
        const dim3 grid(((n_tokens + 31) / 32 + 3) / 4, batch * heads), block(256);
        if (ctx)
            attn_fwd_kernel<true, 128><<<grid, block, 0, s>>>(q, k, vt, ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
        else
            attn_fwd_kernel<false, 128><<<grid, block, 0, s>>>(q, k, vt, ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
        return hipGetLastError();
    }
    if (n_tokens <= 256) {
        const dim3 grid(batch * heads), block(512);
        if (ctx)
            attn_small_kernel<true><<<grid, block, 0, s>>>(q, k, vt, ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
        else
            attn_small_kernel<false><<<grid, block, 0, s>>>(q, k, vt, ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
        return hipGetLastError();
    }
    const int qtiles = (n_tokens + 31) / 32;
    const dim3 grid((qtiles + 3) / 4, batch * heads), block(256);
    if (ctx)
        attn_fwd_kernel<true><<<grid, block, 0, s>>>(q, k, vt, ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
    else
        attn_fwd_kernel<false><<<grid, block, 0, s>>>(q, k, vt, ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
template <int HD = 64>
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16 *__restrict__ Q, const bf16 *__restrict__ Kk,
                                                         const float *__restrict__ lse2, float *__restrict__ attn,
                                                         int N, int npad, float scale2) {
    constexpr int NS = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * 4 + wave) * 32;
    if (q0 >= N) return;  // no barriers in this kernel
    const bf16 *Qb = Q + (int64_t)bh * npad * HD;
    const bf16 *Kb = Kk + (int64_t)bh * npad * HD;
    bf16x8 qf[NS];
    {
        const int qrow = min(q0 + r, N - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *(const bf16x8 *)(Qb + (int64_t)qrow * HD + 16 * s + 8 * h);
    }
    float lr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) lr[e] = lse2[(int64_t)bh * N + min(q0 + acc_row32(e, h), N - 1)];
    float *out = attn + (int64_t)bh * N * N;
    const int ktiles = (N + 31) >> 5;
    bf16x8 kf[NS], kn[NS];
    auto loadk = [&](int kt, bf16x8(&dst)[NS]) {
        const int key = min(kt * 32 + r, N - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) dst[s] = *(const bf16x8 *)(Kb + (int64_t)key * HD + 16 * s + 8 * h);
    };
    loadk(0, kf);
    for (int kt = 0; kt < ktiles; ++kt) {
        if (kt + 1 < ktiles) loadk(kt + 1, kn);
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) S = mfma32(qf[s], kf[s], S);  // rows = queries, col (lane) = key
        const int key = kt * 32 + r;
        if (key < N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qrow = q0 + acc_row32(e, h);
                if (qrow < N) out[(int64_t)qrow * N + key] = fast_exp2(S[e] * scale2 - lr[e]);
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) kf[s] = kn[s];
    }
}

// ------------------------------------------------------------------------------------------
// OCM_PREC_FP32: the same algorithms on fp32 q / k / V^T with v_mfma_f32_32x32x2_f32 (exact fp32
// products). Lane (r, h) supplies one float per operand per instruction (k index = h); a 16-byte read
// of chunk c + 4h of a 128-byte row-half feeds 4 MFMAs. No key permutation is needed: register e of
// the S^T accumulator holds keys acc_row32(e, 0) and acc_row32(e, 1) on the two lane halves, which
// is exactly one instruction's k pair when p[e] is used as the B operand of O^T += V^T P^T.
// LDS tiles: 64 keys x 64 d fp32 = 128 rows of 128 B (row = 2*key + half); V^T likewise (row = 2*d + half).
// (HD = 128, round 4: a key is four 128-byte rows, V^T has 256 row halves: 32 KiB per tile and operand, 128 KiB of dynamic LDS)
template <bool WANT_O, int HD = 64>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float *__restrict__ Q, const float *__restrict__ Kk,
                                                           const float *__restrict__ Vt, float *__restrict__ ctx,
                                                           float *__restrict__ lse2, int N, int npad, int H,
                                                           float scale2) {
    constexpr int TB = HD * 256, RPK = HD / 32, NHF = HD / 32, NDB = HD / 32, CPT = HD / 16;  // tile bytes; 128-B rows per key; ...
    static_assert(HD == 64 || HD == 128, "head widths built on MFMA");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K[2] | Vt[2], TB bytes each (16 KiB at HD = 64)
    char *Ks = smem, *Vs = smem + 2 * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * 4 + wave) * 32;
    const bool active = q0 < N;
    const float *Qb = Q + (int64_t)bh * npad * HD;
    const float *Kb = Kk + (int64_t)bh * npad * HD;
    const float *Vb = Vt + (int64_t)bh * HD * npad;

    f32x4 qf[NHF][4];  // [32-wide part of d][chunk c]: d = 32*part + 16*h + 4*c + e
    {
        const int qrow = min(q0 + r, N - 1);
#pragma unroll
        for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
            for (int c = 0; c < 4; ++c) qf[hf][c] = *(const f32x4 *)(Qb + (int64_t)qrow * HD + 32 * hf + 16 * h + 4 * c);
    }
    // staging: 16 HD K chunks + 16 HD V^T chunks of 16 B per tile, 256 threads -> HD / 16 each. K: LDS row = RPK * key + part
    // (a key's HD floats are RPK rows of 128 B); V^T: LDS row = 2 * d + half of the 64-key run
    f32x4 rk[CPT], rv[CPT];
    auto issue = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int qd = tid + 256 * i, row = qd >> 3, c = qd & 7;  // LDS row, chunk
            const int key = min(kt * 64 + row / RPK, N - 1);
            rk[i] = *(const f32x4 *)(Kb + (int64_t)key * HD + (row % RPK) * 32 + c * 4);
            const int d = row >> 1, key0 = kt * 64 + (row & 1) * 32 + c * 4;  // V^T row d, 4 keys
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (key0 < npad) {  // npad % 8 == 0 and key0 % 4 == 0: the 16-B load stays inside the row
                v = *(const f32x4 *)(Vb + (int64_t)d * npad + key0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (key0 + e >= N) v[e] = 0.f;  // padding keys: exact zeros
            }
            rv[i] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int qd = tid + 256 * i, row = qd >> 3, c = qd & 7;
            *(f32x4 *)(Ks + buf * TB + lds_off(row, c)) = rk[i];
            *(f32x4 *)(Vs + buf * TB + lds_off(row, c)) = rv[i];
        }
    };

    f32x16 O[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int ntiles = (N + 63) >> 6;
    issue(0);
    commit(0);
    lds_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) issue(kt + 1);
        if (active) {
            const char *Kt = Ks + buf * TB, *Vtile = Vs + buf * TB;
            f32x16 S[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int e = 0; e < 16; ++e) S[sub][e] = 0.f;
#pragma unroll
                for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 a = *(const f32x4 *)(Kt + lds_off(RPK * (sub * 32 + r) + hf, c + 4 * h));
#pragma unroll
                        for (int e = 0; e < 4; ++e) S[sub] = mfma32f(a[e], qf[hf][c][e], S[sub]);
                    }
            }
            if ((kt + 1) * 64 > N) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kt * 64 + sub * 32 + acc_row32(e, h) >= N) S[sub][e] = -INFINITY;
            }
            float mx = fmaxf(S[0][0], S[1][0]);
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(S[0][e], S[1][e]));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx * scale2);
            const float alpha = fast_exp2(m - mn);
            m = mn;
            float ps = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = fast_exp2(fmaf(S[sub][e], scale2, -mn));
                    S[sub][e] = p;
                    ps += p;
                }
            l = fmaf(l, alpha, ps);
            if (WANT_O) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[db][e] *= alpha;
                // registers 4g..4g+3 of S[sub] = keys sub*32 + 8g + 4h + (0..3): one 16-B read of V^T
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chunk = sub * 8 + 2 * g + h;  // 4-key chunk index inside the 64-key row
#pragma unroll
                        for (int db = 0; db < NDB; ++db) {
                            const f32x4 a = *(const f32x4 *)(Vtile + lds_off(2 * (db * 32 + r) + (chunk >> 3), chunk & 7));
#pragma unroll
                            for (int e = 0; e < 4; ++e) O[db] = mfma32f(a[e], S[sub][4 * g + e], O[db]);
                        }
                    }
            }
        }
        if (kt + 1 < ntiles) commit(buf ^ 1);
        lds_barrier();
    }
    if (!active) return;
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < N) {
        if (lse2 && h == 0) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
        if (WANT_O) {
            const float inv = 1.0f / lt;
            const int b = bh / H, head = bh - b * H;
            float *dst = ctx + ((int64_t)b * N + qrow) * (H * HD) + head * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = O[db][4 * g + e] * inv;
                    *(f32x4 *)(dst + db * 32 + 8 * g + 4 * h) = o;
                }
        }
    }
}

template <int HD = 64>
__global__ __launch_bounds__(256) void attn_probs_f32_kernel(const float *__restrict__ Q, const float *__restrict__ Kk,
                                                             const float *__restrict__ lse2, float *__restrict__ attn,
                                                             int N, int npad, float scale2) {
    constexpr int NHF = HD / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * 4 + wave) * 32;
    if (q0 >= N) return;
    const float *Qb = Q + (int64_t)bh * npad * HD;
    const float *Kb = Kk + (int64_t)bh * npad * HD;
    f32x4 qf[NHF][4];
    {
        const int qrow = min(q0 + r, N - 1);
#pragma unroll
        for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
            for (int c = 0; c < 4; ++c) qf[hf][c] = *(const f32x4 *)(Qb + (int64_t)qrow * HD + 32 * hf + 16 * h + 4 * c);
    }
    float lr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) lr[e] = lse2[(int64_t)bh * N + min(q0 + acc_row32(e, h), N - 1)];
    float *out = attn + (int64_t)bh * N * N;
    const int ktiles = (N + 31) >> 5;
    for (int kt = 0; kt < ktiles; ++kt) {
        const int key = min(kt * 32 + r, N - 1);
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int hf = 0; hf < NHF; ++hf)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 kf = *(const f32x4 *)(Kb + (int64_t)key * HD + 32 * hf + 16 * h + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) S = mfma32f(qf[hf][c][e], kf[e], S);  // rows = queries, col (lane) = key
            }
        if (kt * 32 + r < N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qrow = q0 + acc_row32(e, h);
                if (qrow < N) out[(int64_t)qrow * N + kt * 32 + r] = fast_exp2(S[e] * scale2 - lr[e]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// OCM_PREC_BF16X3: the flash kernel on split-bf16 pairs (common.h: sp32). q / k rows are 256 bytes
// [hi d 0..31 | lo d 0..31 | hi d 32..63 | lo d 32..63]; V^T rows are key-contiguous in the same 128-byte groups of
// 32 keys. Every product runs as three bf16 MFMAs (hi*hi + hi*lo + lo*hi), P is split in registers. An LDS tile of
// 64 keys is two images (one per 128-byte group) of [64 rows][128 B] with the usual chunk swizzle, so the
// fragment reads are the conflict-free pattern of the bf16 kernel and staging is a byte copy.
// NW waves (4 or 8) of 32 queries share the K / V^T tiles: 8 waves halve the L2 -> LDS bytes per query at the same
// waves per CU (one 8-wave workgroup instead of two 4-wave ones).
// Register-staged form: kept behind knob 6 = 1 for A/B runs; attn_fwd_x3_dma_kernel below is the one dispatched
// (ViT-S/16 B = 64: 35.0 -> 28.4 us per launch; N = 2305: 463 -> 371 us, same box, alternating runs).
template <bool WANT_O, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_x3_kernel(const char *__restrict__ Q, const char *__restrict__ Kk,
                                                             const char *__restrict__ Vt, char *__restrict__ ctx,
                                                             float *__restrict__ lse2, int N, int npad, int H,
                                                             float scale2, int wt) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 16384];  // K[2] | Vt[2], 16 KiB each
    char *Ks = smem, *Vs = smem + 2 * 16384;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * NW + wave) * 32;
    constexpr int NT = NW * 64, CH = 1024 / NT;  // 16-B chunks per thread per tile and operand
    const bool active = q0 < N;  // wave-uniform
    const char *Qb = Q + (int64_t)bh * npad * 256;
    const char *Kb = Kk + (int64_t)bh * npad * 256;
    const char *Vb = Vt + (int64_t)bh * 64 * npad * 4;

    // Q^T as the B operand: lane (query r, half h) holds Q[q0+r][16s + 8h .. +7], hi and lo halves
    bf16x8 qh[4], ql[4];
    {
        const char *qp = Qb + (int64_t)min(q0 + r, N - 1) * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char *p = qp + (s >> 1) * 128 + ((s & 1) * 16 + 8 * h) * 2;
            qh[s] = *(const bf16x8 *)p;
            ql[s] = *(const bf16x8 *)(p + 64);
        }
    }

    // staging: 1024 K chunks + 1024 V^T chunks of 16 B per tile, NT threads -> CH + CH each
    f32x4 rk[CH], rv[CH];
    auto issue = [&](int kt) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int qd = tid + NT * i, row = qd >> 4, c16 = qd & 15;  // row: key (K) or d (V^T); 16 chunks per row
            const int key = min(kt * 64 + row, N - 1);
            rk[i] = *(const f32x4 *)(Kb + (int64_t)key * 256 + c16 * 16);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int key0 = kt * 64 + (c16 >> 3) * 32 + (c16 & 3) * 8;  // first of the chunk's 8 keys (hi or lo halves)
            if (WANT_O && key0 < N) {  // key0 < N <= npad: the chunk lies inside the row
                v = *(const f32x4 *)(Vb + (int64_t)row * npad * 4 + (kt * 2 + (c16 >> 3)) * 128 + (c16 & 7) * 16);
                if (key0 + 8 > N) {  // keys >= N are padding: force exact zeros (0 * garbage must not be NaN)
                    bf16x8 t = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key0 + e >= N) t[e] = (bf16)0.f;
                    v = __builtin_bit_cast(f32x4, t);
                }
            }
            rv[i] = v;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int qd = tid + NT * i, row = qd >> 4, c16 = qd & 15;
            const int o = buf * 16384 + (c16 >> 3) * 8192 + lds_off(row, c16 & 7);
            *(f32x4 *)(Ks + o) = rk[i];
            *(f32x4 *)(Vs + o) = rv[i];
        }
    };

    f32x16 O[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) O[0][e] = O[1][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int pr = pi_row(r);
    const int ntiles = (N + 63) >> 6;

    issue(0);
    commit(0);
    lds_barrier();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
#if !defined(OCM_ABL) || OCM_ABL != 5  // ablation 5: tile 0 reused, no staging after the prologue
        if (kt + 1 < ntiles) issue(kt + 1);
#endif
#if defined(OCM_ABL) && OCM_ABL == 6  // ablation 6: staging only, no arithmetic
        if (false) {
#else
        if (active) {
#endif
            const char *Kt = Ks + buf * 16384, *Vtile = Vs + buf * 16384;
            // the second 32 keys of the tile are all padding in the last tile of N = 197 / 2305 (5 and 1 valid keys):
            // skipping them leaves every result bit for bit (their p is exp2(-inf) = 0 and their V^T columns are 0)
            const bool two = kt * 64 + 32 < N;  // wave-uniform
            f32x16 S[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int e = 0; e < 16; ++e) S[sub][e] = (sub == 1 && !two) ? -INFINITY : 0.f;
                if (sub == 1 && !two) continue;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const char *kp = Kt + (s >> 1) * 8192 + sub * 32 * 128;
                    const bf16x8 kh = *(const bf16x8 *)(kp + lds_off(pr, (s & 1) * 2 + h));
                    const bf16x8 kl = *(const bf16x8 *)(kp + lds_off(pr, 4 + (s & 1) * 2 + h));
                    S[sub] = mfma32x3(kh, kl, qh[s], ql[s], S[sub]);
                }
            }
            if ((kt + 1) * 64 > N) {  // tail tile: padding keys -> -inf (wave-uniform branch)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kt * 64 + sub * 32 + key_of_reg(e, h) >= N) S[sub][e] = -INFINITY;
            }
            float mx = fmaxf(S[0][0], S[1][0]);
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(S[0][e], S[1][e]));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float alpha = defer_max_update(m, mx * scale2);  // every tile holds at least one valid key: finite
            const float mn = m;
            float ps = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (sub == 1 && !two) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = fast_exp2(fmaf(S[sub][e], scale2, -mn));
                    S[sub][e] = p;
                    ps += p;
                }
            }
            l = fmaf(l, alpha, ps);
            if (WANT_O) {
                if (__any(alpha != 1.0f)) {  // the running max moved somewhere in this wave
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        O[0][e] *= alpha;
                        O[1][e] *= alpha;
                    }
                }
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    if (sub == 1 && !two) continue;
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
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const char *vp = Vtile + sub * 8192 + db * 32 * 128;
                            const bf16x8 vh = *(const bf16x8 *)(vp + lds_off(r, 2 * s2 + h));
                            const bf16x8 vl = *(const bf16x8 *)(vp + lds_off(r, 4 + 2 * s2 + h));
                            O[db] = mfma32x3(vh, vl, ph, pl, O[db]);
                        }
                    }
                }
            }
        }
#if !defined(OCM_ABL) || OCM_ABL != 5
        if (kt + 1 < ntiles) commit(buf ^ 1);
#endif
        lds_barrier();
    }

    if (!active) return;
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < N) {
        if (lse2 && h == 0) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
        if (WANT_O) {
            const float inv = 1.0f / lt;
            const int b = bh / H, head = bh - b * H;
            char *dst = ctx + ((int64_t)b * N + qrow) * (H * 64) * 4 + head * 256;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = O[db][4 * g + e] * inv;
                    bf16x4 oh, ol;
                    split4(o, oh, ol);
                    char *p = dst + db * 128 + (8 * g + 4 * h) * 2;
                    if (wt) {  // write-through: the context rows are consumed by another kernel, not by this one
                        const f32x2 dh = __builtin_bit_cast(f32x2, oh), dl = __builtin_bit_cast(f32x2, ol);
                        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(dh) : "memory");
                        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + 64), "v"(dl) : "memory");
                    } else {
                        *(bf16x4 *)p = oh;
                        *(bf16x4 *)(p + 64) = ol;
                    }
                }
        }
    }
}


#ifdef OCM_GEMM_STAMPS
// development only (make stamps, tools/attn_stamps.py): per-wave timeline of attn_fwd_x3_dma_kernel,
// [workgroup (blockIdx.y * gridDim.x + blockIdx.x) < 1024][wave < 8][point < 16]
__device__ unsigned long long g_astamps[1024 * 8 * 16];
extern "C" int ocm_debug_stamps_attn(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_astamps), (size_t)n * 8);
}
#define ASTAMP(k)                                                                                                    \
    do {                                                                                                             \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                    \
        if ((threadIdx.x & 63) == 0 && wg_ < 1024 && (k) < 16)                                                       \
            g_astamps[(wg_ * 8 + (threadIdx.x >> 6 & 7)) * 16 + (k)] = __builtin_readcyclecounter();                 \
    } while (0)
// the constant 100 MHz counter (the same on every XCD), for the entry spread and the span of the grid
#define ASTAMP_RT(k)                                                                                                 \
    do {                                                                                                             \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                    \
        if ((threadIdx.x & 63) == 0 && wg_ < 1024)                                                                   \
            g_astamps[(wg_ * 8 + (threadIdx.x >> 6 & 7)) * 16 + (k)] = wall_clock64();                               \
    } while (0)
#else
#define ASTAMP(k) ((void)0)
#define ASTAMP_RT(k) ((void)0)
#endif

// LDS-DMA form of attn_fwd_x3_kernel (the dispatched one). Same arithmetic per key, but
//   * K / V^T tiles of 32 keys go global -> LDS by `buffer_load ... lds` (1 KiB = 8 LDS rows per wave instruction, the
//     chunk swizzle applied to the per-lane SOURCE offset, the tile index in the scalar offset): no staging registers,
//     no ds_write, no per-tile address arithmetic (the register-staged kernel spent as many vector instructions on
//     staging as on the softmax);
//   * a ring of three 16 KiB stages (K: two [32 keys][128 B] images, V^T: one [64 d][128 B] image), two tiles in
//     flight, one counted `s_waitcnt vmcnt` + one barrier per tile;
//   * 48 KiB of LDS and <= 168 registers: three 4-wave workgroups per CU, so the 768 workgroups of ViT-S/16 at B = 64
//     are resident at once (the 64 KiB / 198-register kernel ran them in one and a half rounds of two).
// Padding keys of the last tile: their K rows are inside the buffer (scores overwritten with -inf), their V^T columns
// are zeroed in LDS after the tile has landed (the qkv epilogue never writes them; 0 * garbage must not be NaN).
// HD = head width: 64 (the DINO ViTs), or 128 (the reference's SimMIM encoder, model.py:93-103) on a two-stage ring of
// 32 KiB stages (K: four [32 keys][128 B] images, V^T: [128 d][128 B]) with twice the Q fragments and context accumulators.
// KSPLIT: long sequences at small batch (one 384^2 window of ViT-S/8 per call: 60 workgroups walking 73 key tiles each).
// blockIdx.z owns a contiguous range of key tiles and leaves its UNNORMALISED context rows, running maximum and sum in `part`
// ([slice][B*H][N][HD + 4] fp32: context, maximum, sum, pad); attn_merge_x3_kernel combines the slices (the flash-decoding reduction).
template <bool WANT_O, int NW, int WPS, int HD = 64, int NSTAGE = 3, bool KSPLIT = false>
__global__ __launch_bounds__(NW * 64, WPS) void attn_fwd_x3_dma_kernel(const char *__restrict__ Q,
                                                                       const char *__restrict__ Kk,
                                                                       const char *__restrict__ Vt, char *__restrict__ ctx,
                                                                       float *__restrict__ lse2, int N, int npad, int H,
                                                                       float scale2, int wt, float *__restrict__ part = nullptr) {
    constexpr int KB = HD * 128, STAGE = 2 * KB, KP = HD / 8 / NW, VP = WANT_O ? HD / 8 / NW : 0, LPS = KP + VP;  // KB: K bytes of a 32-key tile
    constexpr int NQ = HD / 16, ND = HD / 32;  // k slices of the score product, 32-channel blocks of the context
    static_assert((NW == 4 || NW == 8) && (HD == 64 || HD == 128) && (NSTAGE == 2 || NSTAGE == 3), "HD / 8 pieces of 1 KiB per operand and tile");
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    ASTAMP(0);  // entered
    ASTAMP_RT(12);
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * NW + wave) * 32;
    const bool active = q0 < N;  // wave-uniform
    const char *Qb = Q + (int64_t)bh * npad * (HD * 4);
    const char *Kb = Kk + (int64_t)bh * npad * (HD * 4);
    const char *Vb = Vt + (int64_t)bh * HD * npad * 4;
    const int ntiles = (N + 31) >> 5;  // key tiles of the sequence; this workgroup walks nt of them from tile tb
    int tb = 0, nt = ntiles;
    if (KSPLIT) {
        const int per = (ntiles + (int)gridDim.z - 1) / (int)gridDim.z;
        tb = (int)blockIdx.z * per;
        nt = min(per, ntiles - tb);  // >= 1: the launcher sizes gridDim.z so
    }

    // Q^T as the B operand: issued first, so the counted waits below cover it too (vmcnt retires in order)
    bf16x8 qh[NQ], ql[NQ];
    {
        const char *qp = Qb + (int64_t)min(q0 + r, N - 1) * (HD * 4);
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const char *p = qp + (s >> 1) * 128 + ((s & 1) * 16 + 8 * h) * 2;
            qh[s] = *(const bf16x8 *)p;
            ql[s] = *(const bf16x8 *)(p + 64);
        }
    }

    // piece pc = j * NW + wave of a stage: K pieces 0..7 = image (pc >> 2), rows (pc & 3) * 8 .. + 7; V^T pieces = d rows
    int voffK[KP], voffV[VP ? VP : 1];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int pc = j * NW + wave, rho = (pc & 3) * 8 + lrow;
            voffK[j] = rho * (HD * 4) + (pc >> 2) * 128 + ((slot ^ ((rho >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            const int rho = (j * NW + wave) * 8 + lrow;
            voffV[j] = rho * npad * 4 + ((slot ^ ((rho >> 1) & 7)) << 4);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, (unsigned)(npad * HD * 4), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, (unsigned)(HD * npad * 4), 0x00020000);
#define OCM_ATTN_DMA(t, st)                                                                                              \
    do {                                                                                                                 \
        char *st_ = smem + (st) * STAGE;                                                                                 \
        _Pragma("unroll") for (int j = 0; j < KP; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(st_ + (j * NW + wave) * 1024), 16, voffK[j],       \
                                                     (t) * (32 * HD * 4), 0, 0);                                                  \
        _Pragma("unroll") for (int j = 0; j < VP; ++j)                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(st_ + KB + (j * NW + wave) * 1024), 16, voffV[j], \
                                                     (t) * 128, 0, 0);                                                   \
    } while (0)
#else
#define OCM_ATTN_DMA(t, st) (void)0
#endif

    OCM_ATTN_DMA(tb, 0);
    if (NSTAGE == 3 && nt > 1) OCM_ATTN_DMA(tb + 1, 1);
    // Q and the first two tiles have landed before the loop (builtin, not asm: hipcc's own wait-count bookkeeping must
    // see that the Q registers are complete, or it re-waits for them inside the loop with a count that also covers the
    // tile in flight). 0x0F70 = vmcnt(0), expcnt / lgkmcnt untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ASTAMP(1);  // Q and the first two tiles landed

    f32x16 O[ND];
#pragma unroll
    for (int db = 0; db < ND; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int pr = pi_row(r);
    const int first_pad = N - (ntiles - 1) * 32;  // valid keys of the last tile (1..32)
    int sc = 0, si = NSTAGE - 1;                  // stage computed next / filled next

    for (int kt = 0; kt < nt; ++kt) {
        const bool last_of_seq = tb + kt + 1 == ntiles;  // the tile that holds the padding keys
        // own DMAs of tile kt have landed once at most the younger tile kt+1 is pending
        if (NSTAGE == 3 && kt + 1 < nt)
            OCM_VMCNT_ATTN(LPS);
        else
            OCM_VMCNT_ATTN(0);
        // own LDS reads of tile kt-1 retired, then the barrier. Hand-written: a workgroup release fence on LDS makes hipcc
        // wait for vmcnt(0) as well (LDS-DMA writes LDS), which would serialise the tile in flight behind this one
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ASTAMP(2 + kt);  // barrier of tile kt passed
#if !defined(OCM_ABL) || OCM_ABL != 5  // ablation 5: no staging after the prologue
        if (kt + NSTAGE - 1 < nt) OCM_ATTN_DMA(tb + kt + NSTAGE - 1, si);  // into the stage of tile kt-1: everybody is past it
#endif
        char *Kt = smem + sc * STAGE, *Vtile = Kt + KB;
        if (WANT_O && last_of_seq && first_pad < 32) {  // zero the V^T columns of the padding keys (wave-uniform)
            if (tid < 256) {
                const int kc = tid & 3;
                if (kc * 8 + 8 > first_pad) {
#pragma unroll
                    for (int d = tid >> 2; d < HD; d += 64)
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            bf16x8 *p = (bf16x8 *)(Vtile + lds_off(d, half * 4 + kc));
                            bf16x8 t = *p;
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (kc * 8 + e >= first_pad) t[e] = (bf16)0.f;
                            *p = t;
                        }
                }
            }
            lds_barrier();
        }
#if defined(OCM_ABL) && OCM_ABL == 6  // ablation 6: staging only, no arithmetic
        if (false) {
#else
        if (active) {
#endif
            f32x16 S;
#pragma unroll
            for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
            for (int s = 0; s < NQ; ++s) {
                const char *kp = Kt + (s >> 1) * 4096;
                const bf16x8 kh = *(const bf16x8 *)(kp + lds_off(pr, (s & 1) * 2 + h));
                const bf16x8 kl = *(const bf16x8 *)(kp + lds_off(pr, 4 + (s & 1) * 2 + h));
                S = mfma32x3(kh, kl, qh[s], ql[s], S);
            }
            if (last_of_seq && first_pad < 32) {  // padding keys -> -inf (wave-uniform branch)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (key_of_reg(e, h) >= first_pad) S[e] = -INFINITY;
            }
            float mx = S[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mx = fmaxf(mx, S[e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float alpha = defer_max_update(m, mx * scale2);  // every tile holds at least one valid key: finite
            const float mn = m;
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(fmaf(S[e], scale2, -mn));
                S[e] = p;
                ps += p;
            }
            l = fmaf(l, alpha, ps);
            if (WANT_O) {
                if (__any(alpha != 1.0f)) {  // the running max moved somewhere in this wave
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int db = 0; db < ND; ++db) O[db][e] *= alpha;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 ph, pl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float pv = S[8 * s2 + e];
                        const bf16 t = (bf16)pv;
                        ph[e] = t;
                        pl[e] = (bf16)(pv - (float)t);
                    }
#pragma unroll
                    for (int db = 0; db < ND; ++db) {
                        const char *vp = Vtile + db * 32 * 128;
                        const bf16x8 vh = *(const bf16x8 *)(vp + lds_off(r, 2 * s2 + h));
                        const bf16x8 vl = *(const bf16x8 *)(vp + lds_off(r, 4 + 2 * s2 + h));
                        O[db] = mfma32x3(vh, vl, ph, pl, O[db]);
                    }
                }
            }
        }
        sc = sc == NSTAGE - 1 ? 0 : sc + 1;
        si = si == NSTAGE - 1 ? 0 : si + 1;
    }
#undef OCM_ATTN_DMA
    ASTAMP(14);  // key loop done

    if (!active) return;
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (KSPLIT) {
        if (qrow < N) {
            float *dst = part + (((int64_t)blockIdx.z * gridDim.y + bh) * N + qrow) * (HD + 4);
            if (h == 0) *(f32x2 *)(dst + HD) = f32x2{m, lt};
            if (WANT_O) {
#pragma unroll
                for (int db = 0; db < ND; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = O[db][4 * g + e];
                        *(f32x4 *)(dst + 32 * db + 8 * g + 4 * h) = o;
                    }
            }
        }
        return;
    }
    if (qrow < N) {
        if (lse2 && h == 0) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
        if (WANT_O) {
            const float inv = 1.0f / lt;
            const int b = bh / H, head = bh - b * H;
            char *dst = ctx + ((int64_t)b * N + qrow) * (H * HD) * 4 + head * (HD * 4);
#pragma unroll
            for (int db = 0; db < ND; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = O[db][4 * g + e] * inv;
                    bf16x4 oh, ol;
                    split4(o, oh, ol);
                    char *p = dst + db * 128 + (8 * g + 4 * h) * 2;
                    if (wt) {  // write-through: the context rows are consumed by another kernel, not by this one
                        const f32x2 dh = __builtin_bit_cast(f32x2, oh), dl = __builtin_bit_cast(f32x2, ol);
                        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(dh) : "memory");
                        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + 64), "v"(dl) : "memory");
                    } else {
                        *(bf16x4 *)p = oh;
                        *(bf16x4 *)(p + 64) = ol;
                    }
                }
        }
    }
    ASTAMP(15);  // context stores issued
    ASTAMP_RT(13);
}


// ------------------------------------------------------------------------------------------
// Software-pipelined form of attn_fwd_x3_dma_kernel for four-wave workgroups (64-wide heads, whole key range, N <= 1024:
// ViT-S/16 and ViT-B/16 tiles). The loop of the kernel above runs QK^T(t) -> softmax(t) -> P.V(t) per key tile: twelve matrix
// instructions, ~135 vector instructions, twelve matrix instructions, each group waiting for the one before. Here an
// iteration is two BLOCKS,
//     M(t): O += V^T(t) . P(t),  then  S = K(t+1) . Q^T      24 matrix instructions back to back, LDS reads only
//     V(t+1): softmax of S -> P(t+1) (split pairs), running max / sum, rescale of O        vector instructions only
// so a wave alternates between one long matrix burst and one long vector burst, and the three workgroups a CU holds (not
// synchronised with each other: three waves per SIMD) fill each other's bursts: the matrix pipe and the vector ALU are
// separate. ViT-S/16 at B = 64, in the forward, alternating runs on one box: 28.4 -> 27.1 us per launch (stand-alone 29.5 ->
// 28.2); ViT-B/16 at 384^2 the same either way.
// LDS: a ring of three 8 KiB K tiles and a ring of three 8 KiB V^T tiles (M(t) reads V^T(t) and K(t+1)). Batch t of LDS-DMA
// = {K(t+3), V^T(t+2)} is issued right after the barrier that opens iteration t — into the slots of K(t) and V^T(t-1), whose
// last readers finished before that barrier — and waited for (counted vmcnt: everything but batch t) in front of the barrier
// that closes iteration t, one whole iteration before its first reader. One barrier per tile, as before.
// The context rows leave as 16-byte stores: a lane holds four consecutive channels per accumulator group, v_permlane32_swap
// between the two lane halves (same query) makes them eight.
// Same arithmetic per key and the same order of accumulation as attn_fwd_x3_dma_kernel: bit-identical results
// (tools/ab_attn.py checks every shape class with NaN-poisoned padding).
// Measured and NOT shipped (round 4, git history has both): (i) the same blocks with EIGHT waves whose halves run them in
// opposite phases (waves 0-3 in M while waves 4-7 are in V, a barrier between blocks): S, P, O and Q^T live at once need 209-237
// registers, i.e. ONE workgroup per CU where the kernel above (128 registers) runs two — N = 2305, 21 windows: 502 -> 683-787 us
// per launch; forced into 128 registers it spills 105-276 of them. (ii) The kernel above with waves 4-7 half a tile behind
// waves 0-3 (two barriers per tile, 128 registers, two workgroups per CU): 503 -> 527 us. Two independent workgroups per CU
// already interleave the way a ping-pong would; the extra barrier only costs.
template <bool WANT_O>
__global__ __launch_bounds__(256, 3) void attn_fwd_x3_pp_kernel(const char *__restrict__ Q, const char *__restrict__ Kk,
                                                               const char *__restrict__ Vt, char *__restrict__ ctx,
                                                               float *__restrict__ lse2, int N, int npad, int H, float scale2) {
    constexpr int NW = 4, HD = 64, KB = HD * 128, KP = 8 / NW, VP = WANT_O ? 8 / NW : 0;  // KB: bytes of a 32-key tile of K (or V^T)
    __shared__ __attribute__((aligned(1024))) char smem[6 * KB];  // K slots 0..2 | V^T slots 0..2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * NW + wave) * 32;
    const bool active = q0 < N;  // wave-uniform
    const char *Qb = Q + (int64_t)bh * npad * (HD * 4);
    const char *Kb = Kk + (int64_t)bh * npad * (HD * 4);
    const char *Vb = Vt + (int64_t)bh * HD * npad * 4;
    const int nt = (N + 31) >> 5;

    bf16x8 qh[4], ql[4];  // Q^T as the B operand (issued first: the prologue's wait covers it)
    {
        const char *qp = Qb + (int64_t)min(q0 + r, N - 1) * (HD * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char *p = qp + (s >> 1) * 128 + ((s & 1) * 16 + 8 * h) * 2;
            qh[s] = *(const bf16x8 *)p;
            ql[s] = *(const bf16x8 *)(p + 64);
        }
    }
    int voffK[KP], voffV[VP ? VP : 1];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int pc = j * NW + wave, rho = (pc & 3) * 8 + lrow;
            voffK[j] = rho * (HD * 4) + (pc >> 2) * 128 + ((slot ^ ((rho >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            const int rho = (j * NW + wave) * 8 + lrow;
            voffV[j] = rho * npad * 4 + ((slot ^ ((rho >> 1) & 7)) << 4);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, (unsigned)(npad * HD * 4), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, (unsigned)(HD * npad * 4), 0x00020000);
#define OCM_PP_DMA_K(t, slot)                                                                                        \
    do {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < KP; ++j)                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(smem + (slot) * KB + (j * NW + wave) * 1024), 16, \
                                                     voffK[j], (t) * (32 * HD * 4), 0, 0);                           \
    } while (0)
#define OCM_PP_DMA_V(t, slot)                                                                                              \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < VP; ++j)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(smem + (3 + (slot)) * KB + (j * NW + wave) * 1024), 16, \
                                                     voffV[j], (t) * 128, 0, 0);                                           \
    } while (0)
#else
#define OCM_PP_DMA_K(t, slot) (void)0
#define OCM_PP_DMA_V(t, slot) (void)0
#endif
    // prologue: K(0..2), V^T(0..1) — the batches -3 .. -1
    OCM_PP_DMA_K(0, 0);
    if (nt > 1) OCM_PP_DMA_K(1, 1);
    if (WANT_O) OCM_PP_DMA_V(0, 0);
    if (nt > 2) OCM_PP_DMA_K(2, 2);
    if (WANT_O && nt > 1) OCM_PP_DMA_V(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) through the builtin: hipcc must know the Q registers are complete
    // a workgroup barrier that leaves the vector-memory counter alone (LDS-DMA stays in flight across it) and that nothing is
    // scheduled across
    auto block_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    block_barrier();  // everybody's prologue pieces have landed

    f32x16 O[2], S;
    bf16x8 ph[2], pl[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) O[0][e] = O[1][e] = S[e] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 8; ++e) ph[s2][e] = pl[s2][e] = (bf16)0.f;
    float m = -INFINITY, l = 0.f;
    const int pr = pi_row(r);
    const int first_pad = N - (nt - 1) * 32;  // valid keys of the last tile (1..32)

    // S = K(tile in K slot `ks`) . Q^T
    auto qk = [&](int ks) {
        const char *Kt = smem + ks * KB;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char *kp = Kt + (s >> 1) * 4096;
            const bf16x8 kh = *(const bf16x8 *)(kp + lds_off(pr, (s & 1) * 2 + h));
            const bf16x8 kl = *(const bf16x8 *)(kp + lds_off(pr, 4 + (s & 1) * 2 + h));
            S = mfma32x3(kh, kl, qh[s], ql[s], S);
        }
    };
    // block M(t): P.V of tile t (V^T slot vs), then the scores of tile t + 1 (K slot ks) — P's registers are free for them
    auto Mblk = [&](int vs, int ks, bool more) {
        if (WANT_O) {
            const char *Vtile = smem + (3 + vs) * KB;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char *vp = Vtile + db * 32 * 128;
                    const bf16x8 vh = *(const bf16x8 *)(vp + lds_off(r, 2 * s2 + h));
                    const bf16x8 vl = *(const bf16x8 *)(vp + lds_off(r, 4 + 2 * s2 + h));
                    O[db] = mfma32x3(vh, vl, ph[s2], pl[s2], O[db]);
                }
        }
        if (more) qk(ks);
    };
    // block V(t): softmax of the scores in S (tile t), P as split pairs
    auto Vblk = [&](bool last_of_seq) {
        if (last_of_seq && first_pad < 32) {  // padding keys -> -inf (wave-uniform branch)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (key_of_reg(e, h) >= first_pad) S[e] = -INFINITY;
        }
        float mx = S[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, S[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float alpha = defer_max_update(m, mx * scale2);  // every tile holds at least one valid key: finite
        const float mn = m;
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = fast_exp2(fmaf(S[e], scale2, -mn));
            S[e] = p;
            ps += p;
        }
        l = fmaf(l, alpha, ps);
        if (WANT_O) {
            if (__any(alpha != 1.0f)) {  // the running max moved somewhere in this wave
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    O[0][e] *= alpha;
                    O[1][e] *= alpha;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = S[8 * s2 + e];
                    const bf16 t = (bf16)pv;
                    ph[s2][e] = t;
                    pl[s2][e] = (bf16)(pv - (float)t);
                }
        }
    };

    if (active) {
        qk(0);
        Vblk(nt == 1);
    }
    // the first batch of the loop (K(3) into the slot of K(0)) must stay behind everybody's prologue scores
    block_barrier();

    int s0 = 0, s1 = 1;  // slot of tile kt / kt + 1 (both rings: tile t lives in slot t % 3)
    for (int kt = 0; kt < nt; ++kt) {
        const int s2n = s1 == 2 ? 0 : s1 + 1;  // slot of tile kt + 2 (= the slot tile kt - 1 had)
        // batch kt: K(kt+3) into the slot of K(kt), V^T(kt+2) into the slot of V^T(kt-1): everybody is past their readers
        if (kt + 3 < nt) OCM_PP_DMA_K(kt + 3, s0);
        if (WANT_O && kt + 2 < nt) OCM_PP_DMA_V(kt + 2, s2n);
        const bool last = kt + 1 == nt;
        if (WANT_O && last && first_pad < 32) {
            // zero the V^T columns of the padding keys of the last tile (it has landed: waited for in front of the previous
            // barrier; the qkv epilogue never writes them and 0 * garbage must not be NaN)
            const int kc = tid & 3;
            if (kc * 8 + 8 > first_pad) {
                char *Vtile = smem + (3 + s0) * KB;
                const int d = tid >> 2;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    bf16x8 *p = (bf16x8 *)(Vtile + lds_off(d, half * 4 + kc));
                    bf16x8 t = *p;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (kc * 8 + e >= first_pad) t[e] = (bf16)0.f;
                    *p = t;
                }
            }
            block_barrier();
        }
        if (active) {
            Mblk(s0, s1, !last);
            if (!last) Vblk(kt + 2 == nt);
        }
        // everything but batch kt has landed (this wave's pieces; the barrier makes it everybody's)
        if (kt + 3 < nt)
            OCM_VMCNT_ATTN(KP + VP);
        else if (WANT_O && kt + 2 < nt)
            OCM_VMCNT_ATTN(VP);
        else
            OCM_VMCNT_ATTN(0);
        block_barrier();
        s0 = s1;
        s1 = s2n;
    }
#undef OCM_PP_DMA_K
#undef OCM_PP_DMA_V

    if (!active) return;
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (lse2 && h == 0 && qrow < N) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
    if (WANT_O) {
        const float inv = 1.0f / lt;
        const int b = bh / H, head = bh - b * H;
        char *dst = ctx + ((int64_t)b * N + min(qrow, N - 1)) * (H * HD) * 4 + head * (HD * 4);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                f32x4 oa, ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    oa[e] = O[db][4 * g + e] * inv;
                    ob[e] = O[db][4 * g + 4 + e] * inv;
                }
                bf16x4 ah, al, bhh, bl;
                split4(oa, ah, al);
                split4(ob, bhh, bl);
                uint2 ahu = __builtin_bit_cast(uint2, ah), alu = __builtin_bit_cast(uint2, al);
                uint2 bhu = __builtin_bit_cast(uint2, bhh), blu = __builtin_bit_cast(uint2, bl);
                // lanes 32-63 of the first operand swap with lanes 0-31 of the second: the lower lanes end with channels
                // 8g .. 8g+7 (own | partner's), the upper lanes with 8g+8 .. 8g+15 (partner's | own)
                auto sw = [](unsigned &x, unsigned &y) {
                    const auto rr = __builtin_amdgcn_permlane32_swap(x, y, false, false);
                    x = rr[0];
                    y = rr[1];
                };
                sw(ahu.x, bhu.x);
                sw(ahu.y, bhu.y);
                sw(alu.x, blu.x);
                sw(alu.y, blu.y);
                if (qrow < N) {
                    char *p = dst + db * 128 + (8 * g + 8 * h) * 2;
                    *(uint4 *)p = make_uint4(ahu.x, ahu.y, bhu.x, bhu.y);
                    *(uint4 *)(p + 64) = make_uint4(alu.x, alu.y, blu.x, blu.y);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// Wave-split form for a handful of workgroups (one tile per call: ViT-S/16 at B = 1 is 12 workgroups of the kernel above, each
// walking its seven key tiles one barrier at a time on one wave per SIMD — 11.5 us for 0.1 us of matrix work). Here a workgroup
// is ONE 32-query tile of one (image, head) and its four waves split the KEYS: wave w takes key tiles w, w + 4, ... into a
// private two-slot LDS ring (no workgroup barrier in the loop: a wave waits for its own LDS-DMA only), runs the usual
// scores -> softmax -> P.V on them and leaves (m, l, O) in LDS; after the one barrier the four partial results are merged in a
// fixed order (m = max m_w, L = sum l_w 2^(m_w - m), ctx = sum O_w 2^(m_w - m) / L: the flash-decoding reduction of
// attn_merge_x3_kernel inside the workgroup), each wave finishing sixteen channels of the 32 queries as 16-byte pair stores.
// 42 workgroups of at most two tiles per wave instead of 12 of seven. Chosen by launch_attention for N <= 1024 when the kernel
// above would start fewer than 128 workgroups.
template <bool WANT_O>
__global__ __launch_bounds__(256) void attn_fwd_x3_ws_kernel(const char *__restrict__ Q, const char *__restrict__ Kk,
                                                            const char *__restrict__ Vt, char *__restrict__ ctx,
                                                            float *__restrict__ lse2, int N, int npad, int H, float scale2) {
    constexpr int HD = 64, KB = HD * 128, SLOT = 2 * KB;  // a private slot: K tile | V^T tile
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // 4 waves x 2 slots x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int qt = blockIdx.x, bh = blockIdx.y;
    const int q0 = qt * 32;
    const char *Qb = Q + (int64_t)bh * npad * (HD * 4);
    const char *Kb = Kk + (int64_t)bh * npad * (HD * 4);
    const char *Vb = Vt + (int64_t)bh * HD * npad * 4;
    const int nt = (N + 31) >> 5;
    char *mine = smem + wave * 2 * SLOT;

    bf16x8 qh[4], ql[4];
    {
        const char *qp = Qb + (int64_t)min(q0 + r, N - 1) * (HD * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char *p = qp + (s >> 1) * 128 + ((s & 1) * 16 + 8 * h) * 2;
            qh[s] = *(const bf16x8 *)p;
            ql[s] = *(const bf16x8 *)(p + 64);
        }
    }
    // one wave fills a whole tile: pieces 0..7 of K (image pc >> 2, rows (pc & 3) * 8 ..) and of V^T (d rows pc * 8 ..)
    int voffK[8], voffV[WANT_O ? 8 : 1];
    {
        const int lrow = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int rho = (pc & 3) * 8 + lrow;
            voffK[pc] = rho * (HD * 4) + (pc >> 2) * 128 + ((slot ^ ((rho >> 1) & 7)) << 4);
            if (WANT_O) {
                const int rv = pc * 8 + lrow;
                voffV[pc] = rv * npad * 4 + ((slot ^ ((rv >> 1) & 7)) << 4);
            }
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, (unsigned)(npad * HD * 4), 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, (unsigned)(HD * npad * 4), 0x00020000);
#define OCM_WS_DMA(t, sl)                                                                                              \
    do {                                                                                                               \
        char *st_ = mine + (sl) * SLOT;                                                                                \
        _Pragma("unroll") for (int pc = 0; pc < 8; ++pc)                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(st_ + pc * 1024), 16, voffK[pc], (t) * (32 * HD * 4), 0, 0); \
        if (WANT_O) {                                                                                                  \
            _Pragma("unroll") for (int pc = 0; pc < 8; ++pc)                                                           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(st_ + KB + pc * 1024), 16, voffV[pc], (t) * 128, 0, 0); \
        }                                                                                                              \
    } while (0)
#else
#define OCM_WS_DMA(t, sl) (void)0
#endif
    constexpr int LPT = WANT_O ? 16 : 8;  // LDS-DMA instructions per tile
    const int mynt = wave < nt ? (nt - wave + 3) / 4 : 0;  // my tiles: wave, wave + 4, ...
    if (mynt > 0) OCM_WS_DMA(wave, 0);
    if (mynt > 1) OCM_WS_DMA(wave + 4, 1);

    f32x16 O[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) O[0][e] = O[1][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int pr = pi_row(r);
    const int first_pad = N - (nt - 1) * 32;
    for (int i = 0; i < mynt; ++i) {
        const int t = wave + 4 * i, sl = i & 1;
        // my own DMAs of tile i have landed once at most the next tile's are pending (the builtin for the first: it also
        // covers the Q registers, and hipcc must know they are complete)
        if (i == 0) {
            // s_waitcnt immediate (gfx9): vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait"
            if (mynt > 1)
                __builtin_amdgcn_s_waitcnt(0x0F70 | (LPT & 15) | ((LPT >> 4) << 14));  // vmcnt(LPT)
            else
                __builtin_amdgcn_s_waitcnt(0x0F70);
        } else if (i + 1 < mynt) {
            OCM_VMCNT_ATTN(LPT);
        } else {
            OCM_VMCNT_ATTN(0);
        }
        char *Kt = mine + sl * SLOT, *Vtile = Kt + KB;
        const bool last_of_seq = t + 1 == nt;
        if (WANT_O && last_of_seq && first_pad < 32) {  // zero the V^T columns of the padding keys (my own slot: no barrier)
            const int kc = lane & 3;
            if (kc * 8 + 8 > first_pad) {
#pragma unroll
                for (int d = lane >> 2; d < HD; d += 16)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        bf16x8 *p = (bf16x8 *)(Vtile + lds_off(d, half * 4 + kc));
                        bf16x8 tv = *p;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (kc * 8 + e >= first_pad) tv[e] = (bf16)0.f;
                        *p = tv;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char *kp = Kt + (s >> 1) * 4096;
            const bf16x8 kh = *(const bf16x8 *)(kp + lds_off(pr, (s & 1) * 2 + h));
            const bf16x8 kl = *(const bf16x8 *)(kp + lds_off(pr, 4 + (s & 1) * 2 + h));
            S = mfma32x3(kh, kl, qh[s], ql[s], S);
        }
        if (last_of_seq && first_pad < 32) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (key_of_reg(e, h) >= first_pad) S[e] = -INFINITY;
        }
        float mx = S[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, S[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float alpha = defer_max_update(m, mx * scale2);
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = fast_exp2(fmaf(S[e], scale2, -m));
            S[e] = p;
            ps += p;
        }
        l = fmaf(l, alpha, ps);
        if (WANT_O) {
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    O[0][e] *= alpha;
                    O[1][e] *= alpha;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = S[8 * s2 + e];
                    const bf16 tb = (bf16)pv;
                    ph[e] = tb;
                    pl[e] = (bf16)(pv - (float)tb);
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char *vp = Vtile + db * 32 * 128;
                    const bf16x8 vh = *(const bf16x8 *)(vp + lds_off(r, 2 * s2 + h));
                    const bf16x8 vl = *(const bf16x8 *)(vp + lds_off(r, 4 + 2 * s2 + h));
                    O[db] = mfma32x3(vh, vl, ph, pl, O[db]);
                }
            }
        }
        if (i + 2 < mynt) {  // refill this slot: my reads of it are done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            OCM_WS_DMA(t + 8, sl);
        }
    }
#undef OCM_WS_DMA
    // publish (m, l, O) — in my own region, over my slots (their last reads are done) — then merge
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float *Om = (float *)mine;           // [64 d][32 q]
    float *ml = (float *)(mine + 8192);  // m[32] | l[32]
    const float lt = l + __shfl_xor(l, 32, 64);
    if (h == 0) {
        ml[r] = m;
        ml[32 + r] = lt;
    }
    if (WANT_O) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) Om[(32 * db + acc_row32(e, h)) * 32 + r] = O[db][e];
    }
    __syncthreads();
    const int qrow = q0 + r;
    float mw[4], lw[4], mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float *mlw = (const float *)(smem + w * 2 * SLOT + 8192);
        mw[w] = mlw[r];
        lw[w] = mlw[32 + r];
        mm = fmaxf(mm, mw[w]);
    }
    float L = 0.f, wgt[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        wgt[w] = fast_exp2(mw[w] - mm);  // a wave without tiles left m = -inf, l = 0: weight 0
        L = fmaf(lw[w], wgt[w], L);
    }
    if (lse2 && wave == 0 && h == 0 && qrow < N) lse2[(int64_t)bh * N + qrow] = mm + __log2f(L);
    if (WANT_O) {
        const float inv = 1.0f / L;
        const int d0 = 16 * wave + 8 * h;  // this lane finishes channels d0 .. d0 + 7 of query r
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float *Ow = (const float *)(smem + w * 2 * SLOT);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o0[j] = fmaf(Ow[(d0 + j) * 32 + r], wgt[w], o0[j]);
                o1[j] = fmaf(Ow[(d0 + 4 + j) * 32 + r], wgt[w], o1[j]);
            }
        }
        if (qrow < N) {
            bf16x8 hi, lo;
            split8(o0 * inv, o1 * inv, hi, lo);
            const int b = bh / H, head = bh - b * H;
            char *dst = ctx + ((int64_t)b * N + qrow) * (H * HD) * 4 + head * (HD * 4) + (d0 >> 5) * 128 + (d0 & 31) * 2;
            *(bf16x8 *)dst = hi;
            *(bf16x8 *)(dst + 64) = lo;
        }
    }
}

// Combines the key slices of attn_fwd_x3_dma_kernel<..., KSPLIT>: 16 lanes per query row (four channels each),
//   m = max_s m_s,  L = sum_s l_s 2^(m_s - m),  ctx = sum_s O_s 2^(m_s - m) / L  (as split pairs),  lse2 = m + log2 L.
template <int HD>
__global__ __launch_bounds__(256) void attn_merge_x3_kernel(const float *__restrict__ part, int nslice, char *__restrict__ ctx,
                                                            float *__restrict__ lse2, int BH, int N, int H) {
    constexpr int LPR = HD / 4;  // lanes per row
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;  // bh * N + query
    if (row >= (int64_t)BH * N) return;
    const int c = (threadIdx.x % LPR) * 4;
    const int64_t stride = (int64_t)BH * N * (HD + 4);
    const float *p = part + row * (HD + 4);
    float m = -INFINITY;
    for (int s = 0; s < nslice; ++s) m = fmaxf(m, p[s * stride + HD]);
    float L = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nslice; ++s) {
        const f32x2 ml = *(const f32x2 *)(p + s * stride + HD);
        const float wgt = fast_exp2(ml[0] - m);
        L = fmaf(ml[1], wgt, L);
        if (ctx) o += *(const f32x4 *)(p + s * stride + c) * wgt;
    }
    if (ctx) {
        const float inv = 1.0f / L;
        const int bh = (int)(row / N), q = (int)(row - (int64_t)bh * N), b = bh / H, head = bh - b * H;
        bf16x4 hi, lo;
        split4(o * inv, hi, lo);
        char *dst = ctx + ((int64_t)b * N + q) * (H * HD) * 4 + head * (HD * 4) + sp_off(c);
        *(bf16x4 *)dst = hi;
        *(bf16x4 *)(dst + 64) = lo;
    }
    if (lse2 && c == 0) lse2[row] = m + __log2f(L);
}

// Whole-sequence variant of attn_fwd_x3_kernel for N <= 256 (ViT-S/16 and ViT-B/16 at 224^2: N = 197): one workgroup
// of 8 waves per (batch, head). ALL K and V^T tiles of the head (hi + lo halves: up to 128 KiB) are brought into LDS
// by one burst of `buffer_load ... lds` DMAs (no staging registers, the chunk swizzle applied on the source address),
// the V^T columns of padding keys are zeroed in LDS (0 * garbage must not reach the accumulator; the buffer's padding
// columns are never written by the qkv epilogue), then every wave walks the key tiles of its 32 query rows with no
// further synchronisation. K / V^T are read once per head instead of once per 128 queries and the per-tile
// load -> barrier chain of the streaming kernel is gone.
template <bool WANT_O>
__global__ __launch_bounds__(512, 2) void attn_small_x3_kernel(const char *__restrict__ Q, const char *__restrict__ Kk,
                                                               const char *__restrict__ Vt, char *__restrict__ ctx,
                                                               float *__restrict__ lse2, int N, int npad, int H,
                                                               float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntiles = (N + 63) >> 6;
    char *Ks = smem, *Vs = smem + ntiles * 16384;  // per tile: 2 images x 64 rows x 128 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bh = xcd_remap(blockIdx.x, gridDim.x);
    const int q0 = wave * 32;
    const char *Qb = Q + (int64_t)bh * npad * 256;
    const char *Kb = Kk + (int64_t)bh * npad * 256;
    const char *Vb = Vt + (int64_t)bh * 64 * npad * 4;

    // ---- bulk fill: piece = 1 KiB = 8 LDS rows; K tile kt, image g holds rows (keys) of group g of the head row
#if defined(__HIP_DEVICE_COMPILE__)
    {
        typedef __attribute__((address_space(3))) void *lds_ptr;
        const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, (unsigned)(npad * 256), 0x00020000);
        const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, (unsigned)(64 * npad * 4), 0x00020000);
        const int lrow = lane >> 3, slot = lane & 7;
        // K: ntiles * 2 images * 8 pieces; V^T: the same count. Waves take pieces round-robin.
        for (int pc = wave; pc < ntiles * 16; pc += 8) {
            const int kt = pc >> 4, g = (pc >> 3) & 1, rho = (pc & 7) * 8 + lrow;  // row inside the 64-row image
            const int c = slot ^ ((rho >> 1) & 7);
            const int key = min(kt * 64 + rho, N - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(Ks + pc * 1024), 16, key * 256 + g * 128 + c * 16, 0, 0, 0);
            // V^T: row rho = d, the tile's key group g; groups past the padded row length read as zeros (bounds check)
            const int grp = kt * 2 + g;
            const int voff = grp * 128 < npad * 4 ? rho * npad * 4 + grp * 128 + c * 16 : 0x7FFFFFF0;
            if (WANT_O)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(Vs + pc * 1024), 16, voff, 0, 0, 0);
        }
    }
#endif
    // Q^T as the B operand (overlaps the DMA)
    bf16x8 qh[4], ql[4];
    {
        const char *qp = Qb + (int64_t)min(q0 + r, N - 1) * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char *p = qp + (s >> 1) * 128 + ((s & 1) * 16 + 8 * h) * 2;
            qh[s] = *(const bf16x8 *)p;
            ql[s] = *(const bf16x8 *)(p + 64);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (WANT_O && (N & 63)) {  // zero the V^T halves of the padding keys of the last tile: keys N .. 64*ntiles - 1
        const int kt = ntiles - 1, first = N - kt * 64;  // first padding key inside the tile
        for (int i = tid; i < 64 * (64 - first); i += 512) {
            const int d = i / (64 - first), key = first + i % (64 - first);
            const int g = key >> 5, kk = key & 31;
            char *base = Vs + kt * 16384 + g * 8192 + (kk & 7) * 2;
            *(bf16 *)(base + lds_off(d, kk >> 3)) = (bf16)0.f;        // hi half: chunks 0..3 of the row
            *(bf16 *)(base + lds_off(d, 4 + (kk >> 3))) = (bf16)0.f;  // lo half: chunks 4..7
        }
    }
    __syncthreads();
    if (q0 >= N) return;  // no barrier below

    f32x16 O[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) O[0][e] = O[1][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int pr = pi_row(r);
    for (int kt = 0; kt < ntiles; ++kt) {
        const char *Kt = Ks + kt * 16384, *Vtile = Vs + kt * 16384;
        f32x16 S[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int e = 0; e < 16; ++e) S[sub][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const char *kp = Kt + (s >> 1) * 8192 + sub * 32 * 128;
                const bf16x8 kh = *(const bf16x8 *)(kp + lds_off(pr, (s & 1) * 2 + h));
                const bf16x8 kl = *(const bf16x8 *)(kp + lds_off(pr, 4 + (s & 1) * 2 + h));
                S[sub] = mfma32x3(kh, kl, qh[s], ql[s], S[sub]);
            }
        }
        if ((kt + 1) * 64 > N) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (kt * 64 + sub * 32 + key_of_reg(e, h) >= N) S[sub][e] = -INFINITY;
        }
        float mx = fmaxf(S[0][0], S[1][0]);
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, fmaxf(S[0][e], S[1][e]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float alpha = defer_max_update(m, mx * scale2);  // every tile holds at least one valid key: finite
        const float mn = m;
        float ps = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = fast_exp2(fmaf(S[sub][e], scale2, -mn));
                S[sub][e] = p;
                ps += p;
            }
        l = fmaf(l, alpha, ps);
        if (WANT_O) {
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    O[0][e] *= alpha;
                    O[1][e] *= alpha;
                }
            }
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
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const char *vp = Vtile + sub * 8192 + db * 32 * 128;
                        const bf16x8 vh = *(const bf16x8 *)(vp + lds_off(r, 2 * s2 + h));
                        const bf16x8 vl = *(const bf16x8 *)(vp + lds_off(r, 4 + 2 * s2 + h));
                        O[db] = mfma32x3(vh, vl, ph, pl, O[db]);
                    }
                }
        }
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    const int qrow = q0 + r;
    if (qrow < N) {
        if (lse2 && h == 0) lse2[(int64_t)bh * N + qrow] = m + __log2f(lt);
        if (WANT_O) {
            const float inv = 1.0f / lt;
            const int b = bh / H, head = bh - b * H;
            char *dst = ctx + ((int64_t)b * N + qrow) * (H * 64) * 4 + head * 256;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = O[db][4 * g + e] * inv;
                    bf16x4 oh, ol;
                    split4(o, oh, ol);
                    char *p = dst + db * 128 + (8 * g + 4 * h) * 2;
                    *(bf16x4 *)p = oh;
                    *(bf16x4 *)(p + 64) = ol;
                }
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_probs_x3_kernel(const char *__restrict__ Q, const char *__restrict__ Kk,
                                                            const float *__restrict__ lse2, float *__restrict__ attn,
                                                            int N, int npad, float scale2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    int qblk, bh;
    xcd_remap2(qblk, bh);
    const int q0 = (qblk * 4 + wave) * 32;
    if (q0 >= N) return;  // no barriers in this kernel
    constexpr int NQ = HD / 16;
    const char *Qb = Q + (int64_t)bh * npad * (HD * 4);
    const char *Kb = Kk + (int64_t)bh * npad * (HD * 4);
    auto loadrow = [&](const char *base, int row, bf16x8(&hi)[NQ], bf16x8(&lo)[NQ]) {
        const char *rp = base + (int64_t)row * (HD * 4);
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const char *p = rp + (s >> 1) * 128 + ((s & 1) * 16 + 8 * h) * 2;
            hi[s] = *(const bf16x8 *)p;
            lo[s] = *(const bf16x8 *)(p + 64);
        }
    };
    bf16x8 qh[NQ], ql[NQ];
    loadrow(Qb, min(q0 + r, N - 1), qh, ql);
    float lr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) lr[e] = lse2[(int64_t)bh * N + min(q0 + acc_row32(e, h), N - 1)];
    float *out = attn + (int64_t)bh * N * N;
    // blockIdx.z: a contiguous share of the key tiles (every probability is independent of the others: small batches are cut
    // along the keys too, so that a one-tile call is more than a dozen workgroups walking the whole row)
    const int nkt = (N + 31) >> 5, per = (nkt + (int)gridDim.z - 1) / (int)gridDim.z;
    const int kt0 = (int)blockIdx.z * per, ktiles = min(nkt, kt0 + per);
    if (kt0 >= ktiles) return;
    bf16x8 kh[NQ], kl[NQ], nh[NQ], nl[NQ];
    loadrow(Kb, min(kt0 * 32 + r, N - 1), kh, kl);
    for (int kt = kt0; kt < ktiles; ++kt) {
        if (kt + 1 < ktiles) loadrow(Kb, min((kt + 1) * 32 + r, N - 1), nh, nl);
        f32x16 S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NQ; ++s) S = mfma32x3(qh[s], ql[s], kh[s], kl[s], S);  // rows = queries, col (lane) = key
        const int key = kt * 32 + r;
        if (key < N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qrow = q0 + acc_row32(e, h);
                if (qrow < N) out[(int64_t)qrow * N + key] = fast_exp2(S[e] * scale2 - lr[e]);
            }
        }
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            kh[s] = nh[s];
            kl[s] = nl[s];
        }
    }
}


// ------------------------------------------------------------------------------------------
// Heads that are not 64 channels wide (the reference's SimMIM pre-training encoder, model.py:93-103: 3 heads of 128):
// plain fp32 FMA attention on the fp32 (3,B,H,N,hd) qkv tensor, one wavefront per query row. Not a hot path —
// correctness-grade for every precision mode (the arithmetic is fp32 whatever the GEMMs' operand type is).
//   scores: lanes over keys (each lane walks one K row), q broadcast from LDS;  softmax: wave reductions;
//   context: lanes over d, p broadcast from LDS, V rows read coalesced.
__device__ __forceinline__ void store_ctx1(int prec, char *rowp, int col, float v) {
    if (prec == 0) {
        *(bf16 *)(rowp + col * 2) = (bf16)v;
    } else if (prec == 1) {
        *(float *)(rowp + col * 4) = v;
    } else {
        bf16 hi, lo;
        split1(v, hi, lo);
        *(bf16 *)(rowp + sp_off(col)) = hi;
        *(bf16 *)(rowp + sp_off(col) + 64) = lo;
    }
}

__global__ __launch_bounds__(256) void attn_generic_kernel(const float *__restrict__ qkv, char *__restrict__ ctx,
                                                           float *__restrict__ attn, const int32_t *__restrict__ query_rows,
                                                           int n_rows, float *__restrict__ rows, int B, int H, int N,
                                                           int hd, float scale, int prec, bool all_queries) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *qs = (float *)smem + wave * (hd + N);  // this wave's query, then its N scores / probabilities
    float *sc = qs + hd;
    const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
    const int qi = blockIdx.x * 4 + wave;
    const int nq = all_queries ? N : n_rows;
    if (qi >= nq) return;  // no workgroup barriers in this kernel
    const int query = all_queries ? qi : (query_rows ? query_rows[qi] : 0);
    const size_t plane = (size_t)B * H * N * hd;
    const float *Q = qkv + ((size_t)bh * N + query) * hd;
    const float *K = qkv + plane + (size_t)bh * N * hd;
    const float *V = qkv + 2 * plane + (size_t)bh * N * hd;
    for (int d = lane; d < hd; d += 64) qs[d] = Q[d];
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int key = lane; key < N; key += 64) {
        const float *kp = K + (size_t)key * hd;
        float acc = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const f32x4 kv = *(const f32x4 *)(kp + d), qv = *(const f32x4 *)(qs + d);
            acc = fmaf(qv[0], kv[0], acc);
            acc = fmaf(qv[1], kv[1], acc);
            acc = fmaf(qv[2], kv[2], acc);
            acc = fmaf(qv[3], kv[3], acc);
        }
        acc *= scale;
        sc[key] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int key = lane; key < N; key += 64) {
        const float p = __expf(sc[key] - mx);
        sc[key] = p;
        sum += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (int key = lane; key < N; key += 64) sc[key] *= inv;
    __builtin_amdgcn_wave_barrier();
    if (attn && all_queries) {
        float *dst = attn + ((size_t)bh * N + query) * N;
        for (int key = lane; key < N; key += 64) dst[key] = sc[key];
    }
    if (rows) {  // selected rows with the CLS column dropped (utils.py:232)
        if (all_queries) {
            for (int r = 0; r < n_rows; ++r)
                if ((query_rows ? query_rows[r] : 0) == query) {
                    float *dst = rows + ((size_t)bh * n_rows + r) * (N - 1);
                    for (int key = 1 + lane; key < N; key += 64) dst[key - 1] = sc[key];
                }
        } else {
            float *dst = rows + ((size_t)bh * n_rows + qi) * (N - 1);
            for (int key = 1 + lane; key < N; key += 64) dst[key - 1] = sc[key];
        }
    }
    if (ctx && all_queries) {
        char *rowp = ctx + ((size_t)b * N + query) * (size_t)(H * hd) * (prec ? 4 : 2);
        for (int d = lane; d < hd; d += 64) {
            float acc = 0.f;
            for (int key = 0; key < N; ++key) acc = fmaf(sc[key], V[(size_t)key * hd + d], acc);
            store_ctx1(prec, rowp, head * hd + d, acc);
        }
    }
}

hipError_t launch_attention_generic(int prec, const float *qkv, void *ctx, float *attn, const int32_t *query_rows,
                                    int n_rows, float *rows, int batch, int n_tokens, int heads, int head_dim, float scale,
                                    hipStream_t s) {
    if (head_dim % 4 || head_dim > 512 || n_tokens > 8192) return hipErrorInvalidValue;
    if (prec == 2 && (heads * head_dim) % 32) return hipErrorInvalidValue;
    const bool all = ctx || attn;
    const int nq = all ? n_tokens : n_rows;
    if (nq <= 0) return hipSuccess;
    const size_t lds = (size_t)4 * (head_dim + n_tokens) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static unsigned long long optin_mask = 0;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (lds > 64 * 1024 && !(optin_mask >> (dev & 63) & 1)) {
        if (hipError_t e = hipFuncSetAttribute((const void *)attn_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            e != hipSuccess)
            return e;
        optin_mask |= 1ull << (dev & 63);
    }
    attn_generic_kernel<<<dim3((nq + 3) / 4, batch * heads), dim3(256), lds, s>>>(
        qkv, (char *)ctx, attn, query_rows, n_rows, rows, batch, heads, n_tokens, head_dim, scale, prec, all);
    return hipGetLastError();
}

size_t attention_ksplit_bytes(int prec, int batch, int n_tokens, int heads, int head_dim) {
    if (prec != 2 || head_dim != 64 || n_tokens <= 1024) return 0;
    const int wgs = (((n_tokens + 31) / 32 + 7) / 8) * batch * heads;
    if (wgs > 128) return 0;
    return (size_t)4 * batch * heads * n_tokens * (64 + 4) * sizeof(float);
}

hipError_t launch_attention(int prec, const void *q, const void *k, const void *vt, void *ctx, float *lse2, int batch,
                            int n_tokens, int n_pad, int heads, float scale, hipStream_t s, int head_dim, float *ksplit_ws,
                            size_t ksplit_bytes) {
    if (head_dim == 128 && prec == 2) {  // 128-wide heads (model.py:93-103): split-bf16 only, four wavefronts, two stages
        if (n_pad % 32) return hipErrorInvalidValue;
        const dim3 g128(((n_tokens + 31) / 32 + 3) / 4, batch * heads), b128(256);
        if (ctx)
            attn_fwd_x3_dma_kernel<true, 4, 2, 128, 2><<<g128, b128, 0, s>>>((const char *)q, (const char *)k, (const char *)vt,
                                                                             (char *)ctx, lse2, n_tokens, n_pad, heads,
                                                                             scale * LOG2E, 0);
        else
            attn_fwd_x3_dma_kernel<false, 4, 2, 128, 2><<<g128, b128, 0, s>>>((const char *)q, (const char *)k, (const char *)vt,
                                                                              (char *)ctx, lse2, n_tokens, n_pad, heads,
                                                                              scale * LOG2E, 0);
        return hipGetLastError();
    }
    if (head_dim != 64 && head_dim != 128) return hipErrorInvalidValue;
    if (!prec)
        return launch_attention_bf16((const bf16 *)q, (const bf16 *)k, (const bf16 *)vt, (bf16 *)ctx, lse2, batch,
                                     n_tokens, n_pad, heads, scale, s, head_dim);
    if (prec == 1 && head_dim == 128) {  // 128-wide heads in fp32 precision: the streaming kernel on 128 KiB of dynamic LDS
        const dim3 g128(((n_tokens + 31) / 32 + 3) / 4, batch * heads), b128(256);
        constexpr int LDSF = 4 * 128 * 256;
        static unsigned long long optin[2] = {0, 0};
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
        const void *kern = ctx ? (const void *)attn_fwd_f32_kernel<true, 128> : (const void *)attn_fwd_f32_kernel<false, 128>;
        unsigned long long &mask = optin[ctx ? 1 : 0];
        if (!(mask >> (dev & 63) & 1)) {
            if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSF); e != hipSuccess) return e;
            mask |= 1ull << (dev & 63);
        }
        if (ctx)
            attn_fwd_f32_kernel<true, 128><<<g128, b128, LDSF, s>>>((const float *)q, (const float *)k, (const float *)vt,
                                                                    (float *)ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
        else
            attn_fwd_f32_kernel<false, 128><<<g128, b128, LDSF, s>>>((const float *)q, (const float *)k, (const float *)vt,
                                                                     (float *)ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
        return hipGetLastError();
    }
    if (head_dim != 64) return hipErrorInvalidValue;
    const int qtiles = (n_tokens + 31) / 32;
    const dim3 grid((qtiles + 3) / 4, batch * heads), block(256);
    if (prec == 2) {
        if (n_pad % 32) return hipErrorInvalidValue;
        // Whole-sequence kernel (all K / V^T of a head in LDS, one DMA burst): built, parity-green, and measured SLOWER
        // than the streaming kernel at ViT-S/16, B = 64 (40.3 us against 34.1 us per launch: 128 KiB of LDS leaves one
        // workgroup per CU, so nothing overlaps the fill) — kept behind knob 6 = 2 for A/B runs, not dispatched.
#ifdef OCM_DEV
        if (n_tokens <= 256 && OCM_KNOB(6) == 2) {
            const int nt = (n_tokens + 63) / 64, lds = nt * 2 * 16384;
            static unsigned long long optin[2] = {0, 0};
            int dev = 0;
            if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
            const void *kern = ctx ? (const void *)attn_small_x3_kernel<true> : (const void *)attn_small_x3_kernel<false>;
            unsigned long long &mask = optin[ctx ? 1 : 0];
            if (!(mask >> (dev & 63) & 1)) {
                if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
                    e != hipSuccess)
                    return e;
                mask |= 1ull << (dev & 63);
            }
            const dim3 g1(batch * heads), b1(512);
            if (ctx)
                attn_small_x3_kernel<true><<<g1, b1, lds, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx,
                                                               lse2, n_tokens, n_pad, heads, scale * LOG2E);
            else
                attn_small_x3_kernel<false><<<g1, b1, lds, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx,
                                                                lse2, n_tokens, n_pad, heads, scale * LOG2E);
            return hipGetLastError();
        }
#endif
        // long sequences: 8 waves per workgroup share each K / V^T tile (half the L2 -> LDS traffic per query)
        // (the dozen workgroups of a one-tile call at N = 197 stay on four waves: eight measured 0.65 -> 0.68 ms per forward)
        const bool wide = (n_tokens > 1024 && OCM_KNOB(7) != 1) || OCM_KNOB(7) == 2;
        const dim3 grid8((qtiles + 7) / 8, batch * heads), block8(512);
        // Long sequence, few workgroups (one ViT-S/8 window per call: 60 of them, 73 key tiles each): cut the key range into
        // up to four slices per workgroup and merge (attn_merge_x3_kernel)
        // (the statistics-only pass of get_last_selfattention takes the same slices, so the log-sum-exp — and with it the returned
        // probabilities — has the same bits from both entry points)
        if (wide && ksplit_ws && OCM_KNOB(7) == 0) {
            const int wgs = (int)(grid8.x * grid8.y), ktiles = (n_tokens + 31) / 32;
            // (eight slices of 9-10 tiles measure slower than four of 18-19: 36.6 + 7.4 us against 34.3 + 5.1 us with the merge)
            const int want = wgs <= 64 ? 4 : wgs <= 96 ? 3 : wgs <= 128 ? 2 : 1;
            const int per = (ktiles + want - 1) / want, nslice = (ktiles + per - 1) / per;  // every slice owns >= 1 tile
            const size_t need = (size_t)nslice * batch * heads * n_tokens * (64 + 4) * sizeof(float);
            if (nslice > 1 && need <= ksplit_bytes) {
                if (ctx)
                    attn_fwd_x3_dma_kernel<true, 8, 2, 64, 3, true><<<dim3(grid8.x, grid8.y, nslice), block8, 0, s>>>(
                        (const char *)q, (const char *)k, (const char *)vt, (char *)ctx, lse2, n_tokens, n_pad, heads,
                        scale * LOG2E, 0, ksplit_ws);
                else
                    attn_fwd_x3_dma_kernel<false, 8, 2, 64, 3, true><<<dim3(grid8.x, grid8.y, nslice), block8, 0, s>>>(
                        (const char *)q, (const char *)k, (const char *)vt, (char *)ctx, lse2, n_tokens, n_pad, heads,
                        scale * LOG2E, 0, ksplit_ws);
                if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
                const int64_t rows = (int64_t)batch * heads * n_tokens;
                attn_merge_x3_kernel<64><<<dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s>>>(
                    ksplit_ws, nslice, (char *)ctx, lse2, batch * heads, n_tokens, heads);
                return hipGetLastError();
            }
        }
#define OCM_X3_ATTN(WO, NW_, G, B_)                                                                                      \
    attn_fwd_x3_kernel<WO, NW_><<<G, B_, 0, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx, lse2, \
                                                 n_tokens, n_pad, heads, scale * LOG2E, (ocm_wt_mask() >> 4) & 1)
#define OCM_X3_ATTN_DMA(WO, NW_, WPS_, G, B_)                                                                          \
    attn_fwd_x3_dma_kernel<WO, NW_, WPS_><<<G, B_, 0, s>>>((const char *)q, (const char *)k, (const char *)vt,          \
                                                           (char *)ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E,    \
                                                           (ocm_wt_mask() >> 4) & 1)
#ifdef OCM_DEV
        if (OCM_KNOB(6) == 1) {  // the register-staged streaming kernel (A/B runs)
            if (wide) {
                if (ctx) OCM_X3_ATTN(true, 8, grid8, block8); else OCM_X3_ATTN(false, 8, grid8, block8);
            } else {
                if (ctx) OCM_X3_ATTN(true, 4, grid, block); else OCM_X3_ATTN(false, 4, grid, block);
            }
        } else
#endif
        if (wide) {
            if (ctx) OCM_X3_ATTN_DMA(true, 8, 2, grid8, block8); else OCM_X3_ATTN_DMA(false, 8, 2, grid8, block8);
        } else if (grid.x * grid.y < 128 && OCM_KNOB(6) != 4 && OCM_KNOB(6) != 3) {
            // a handful of workgroups (one tile per call): the keys of a 32-query tile split over the four waves of a workgroup
            const dim3 gws(qtiles, batch * heads);
            constexpr int LDSWS = 4 * 2 * 2 * 8192;
            static unsigned long long optin[2] = {0, 0};
            int dev = 0;
            if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
            const void *kern = ctx ? (const void *)attn_fwd_x3_ws_kernel<true> : (const void *)attn_fwd_x3_ws_kernel<false>;
            unsigned long long &mask = optin[ctx ? 1 : 0];
            if (!(mask >> (dev & 63) & 1)) {
                if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSWS); e != hipSuccess)
                    return e;
                mask |= 1ull << (dev & 63);
            }
            if (ctx)
                attn_fwd_x3_ws_kernel<true><<<gws, block, LDSWS, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx,
                                                                      lse2, n_tokens, n_pad, heads, scale * LOG2E);
            else
                attn_fwd_x3_ws_kernel<false><<<gws, block, LDSWS, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx,
                                                                       lse2, n_tokens, n_pad, heads, scale * LOG2E);
        } else if (OCM_KNOB(6) == 3) {  // development A/B: the round-3 loop on four waves (shipped: the software-pipelined kernel)
            if (ctx) OCM_X3_ATTN_DMA(true, 4, 3, grid, block); else OCM_X3_ATTN_DMA(false, 4, 3, grid, block);
        } else if (ctx) {
            attn_fwd_x3_pp_kernel<true><<<grid, block, 0, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx, lse2,
                                                               n_tokens, n_pad, heads, scale * LOG2E);
        } else {
            attn_fwd_x3_pp_kernel<false><<<grid, block, 0, s>>>((const char *)q, (const char *)k, (const char *)vt, (char *)ctx, lse2,
                                                                n_tokens, n_pad, heads, scale * LOG2E);
        }
#undef OCM_X3_ATTN_DMA
#undef OCM_X3_ATTN
        return hipGetLastError();
    }
    constexpr int LDSF64 = 4 * 64 * 256;  // 64 KiB: K[2] | Vt[2]
    if (ctx)
        attn_fwd_f32_kernel<true><<<grid, block, LDSF64, s>>>((const float *)q, (const float *)k, (const float *)vt,
                                                              (float *)ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
    else
        attn_fwd_f32_kernel<false><<<grid, block, LDSF64, s>>>((const float *)q, (const float *)k, (const float *)vt,
                                                               (float *)ctx, lse2, n_tokens, n_pad, heads, scale * LOG2E);
    return hipGetLastError();
}

hipError_t launch_attention_probs(int prec, const void *q, const void *k, const float *lse2, float *attn, int batch,
                                  int n_tokens, int n_pad, int heads, float scale, hipStream_t s, int head_dim) {
    const int qtiles = (n_tokens + 31) / 32;
    const dim3 grid((qtiles + 3) / 4, batch * heads), block(256);
    // split-bf16 kernels: key chunks in grid.z until there are ~512 workgroups (a one-tile call is 12 .. 114 otherwise)
    const int wgs = (int)(grid.x * grid.y), ktiles = qtiles;
    int nz = wgs >= 256 ? 1 : min(ktiles, min(16, (512 + wgs - 1) / wgs));
    if (OCM_KNOB(2) > 0) nz = min(ktiles, OCM_KNOB(2));  // development A/B
    const dim3 gridz(grid.x, grid.y, nz);
    if (head_dim == 128 && prec == 2) {
        attn_probs_x3_kernel<128><<<gridz, block, 0, s>>>((const char *)q, (const char *)k, lse2, attn, n_tokens, n_pad,
                                                          scale * LOG2E);
        return hipGetLastError();
    }
    if (head_dim == 128) {  // 128-wide heads in the fp32 / single-bf16 precisions
        if (prec)
            attn_probs_f32_kernel<128><<<grid, block, 0, s>>>((const float *)q, (const float *)k, lse2, attn, n_tokens, n_pad,
                                                              scale * LOG2E);
        else
            attn_probs_kernel<128><<<grid, block, 0, s>>>((const bf16 *)q, (const bf16 *)k, lse2, attn, n_tokens, n_pad,
                                                          scale * LOG2E);
        return hipGetLastError();
    }
    if (head_dim != 64) return hipErrorInvalidValue;
    if (prec == 2)
        attn_probs_x3_kernel<64><<<gridz, block, 0, s>>>((const char *)q, (const char *)k, lse2, attn, n_tokens, n_pad,
                                                     scale * LOG2E);
    else if (prec)
        attn_probs_f32_kernel<64><<<grid, block, 0, s>>>((const float *)q, (const float *)k, lse2, attn, n_tokens, n_pad,
                                                         scale * LOG2E);
    else
        attn_probs_kernel<64><<<grid, block, 0, s>>>((const bf16 *)q, (const bf16 *)k, lse2, attn, n_tokens, n_pad,
                                                     scale * LOG2E);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// rows[b][h][i][j-1] = softmax_j(q[query_i] . k[j] * scale), j = 1..N-1   (utils.py:232)
// elements 8c .. 8c+7 of a head row as fp32
__device__ __forceinline__ void load8(const bf16 *rowp, int c, float (&o)[8]) {
    const bf16x8 t = *(const bf16x8 *)(rowp + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)t[e];
}
__device__ __forceinline__ void load8(const float *rowp, int c, float (&o)[8]) {
    const f32x4 a = *(const f32x4 *)(rowp + c * 8), b = *(const f32x4 *)(rowp + c * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = a[e];
        o[4 + e] = b[e];
    }
}
__device__ __forceinline__ void load8(const sp32 *rowp, int c, float (&o)[8]) {  // hi + lo
    const char *p = (const char *)rowp + sp_off(c * 8);
    const bf16x8 hi = *(const bf16x8 *)p, lo = *(const bf16x8 *)(p + 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)hi[e] + (float)lo[e];
}

template <class E, int HD>
__global__ __launch_bounds__(64) void attn_rows_kernel(const E *__restrict__ Q, const E *__restrict__ Kk,
                                                       const int32_t *__restrict__ query_rows, int n_rows,
                                                       float *__restrict__ rows, int N, int npad, float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sc = (float *)smem;  // N scores
    const int lane = threadIdx.x, bh = blockIdx.y, qi = blockIdx.x;
    const int query = query_rows ? query_rows[qi] : 0;
    const E *qp = Q + ((int64_t)bh * npad + query) * HD;
    float qv[HD];
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
        float t[8];
        load8(qp, c, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[c * 8 + e] = t[e];
    }
    float mx = -INFINITY;
    for (int key = lane; key < N; key += 64) {
        const E *kp = Kk + ((int64_t)bh * npad + key) * HD;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            float t[8];
            load8(kp, c, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qv[c * 8 + e], t[e], acc);
        }
        acc *= scale2;
        sc[key] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int key = lane; key < N; key += 64) {
        const float p = fast_exp2(sc[key] - mx);
        sc[key] = p;
        sum += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    float *dst = rows + ((int64_t)bh * n_rows + qi) * (N - 1);
    for (int key = 1 + lane; key < N; key += 64) dst[key - 1] = sc[key] * inv;
}

// When the full probabilities of the block are materialised anyway, the selected rows are a slice of them:
// rows[bh][qi][j] = attn[bh][query[qi]][1 + j]  (bit-identical to the matrix the caller also receives).
__global__ __launch_bounds__(256) void rows_from_probs_kernel(const float *__restrict__ attn,
                                                              const int32_t *__restrict__ query_rows, int n_rows,
                                                              float *__restrict__ rows, int N, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i % (N - 1));
        const size_t t = i / (N - 1);
        const int qi = (int)(t % n_rows);
        const size_t bh = t / n_rows;
        const int query = query_rows ? query_rows[qi] : 0;
        rows[i] = attn[(bh * N + query) * N + 1 + j];
    }
}

hipError_t launch_rows_from_probs(const float *attn, const int32_t *query_rows, int n_rows, float *rows, int batch,
                                  int n_tokens, int heads, hipStream_t s) {
    const size_t total = (size_t)batch * heads * n_rows * (n_tokens - 1);
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    rows_from_probs_kernel<<<dim3(blocks), dim3(256), 0, s>>>(attn, query_rows, n_rows, rows, n_tokens, total);
    return hipGetLastError();
}

hipError_t launch_attention_rows(int prec, const void *q, const void *k, const int32_t *query_rows, int n_rows,
                                 float *rows, int batch, int n_tokens, int n_pad, int heads, float scale, hipStream_t s,
                                 int head_dim) {
    if (n_rows <= 0) return hipSuccess;
    const dim3 grid(n_rows, batch * heads), block(64);
    const size_t lds = (size_t)n_tokens * sizeof(float);
    if (head_dim == 128) {
        if (prec == 2)
            attn_rows_kernel<sp32, 128><<<grid, block, lds, s>>>((const sp32 *)q, (const sp32 *)k, query_rows, n_rows, rows,
                                                                 n_tokens, n_pad, scale * LOG2E);
        else if (prec)
            attn_rows_kernel<float, 128><<<grid, block, lds, s>>>((const float *)q, (const float *)k, query_rows, n_rows, rows,
                                                                  n_tokens, n_pad, scale * LOG2E);
        else
            attn_rows_kernel<bf16, 128><<<grid, block, lds, s>>>((const bf16 *)q, (const bf16 *)k, query_rows, n_rows, rows,
                                                                 n_tokens, n_pad, scale * LOG2E);
        return hipGetLastError();
    }
    if (head_dim != 64) return hipErrorInvalidValue;
    if (prec == 2)
        attn_rows_kernel<sp32, 64><<<grid, block, lds, s>>>((const sp32 *)q, (const sp32 *)k, query_rows, n_rows, rows,
                                                        n_tokens, n_pad, scale * LOG2E);
    else if (prec)
        attn_rows_kernel<float, 64><<<grid, block, lds, s>>>((const float *)q, (const float *)k, query_rows, n_rows, rows,
                                                         n_tokens, n_pad, scale * LOG2E);
    else
        attn_rows_kernel<bf16, 64><<<grid, block, lds, s>>>((const bf16 *)q, (const bf16 *)k, query_rows, n_rows, rows,
                                                        n_tokens, n_pad, scale * LOG2E);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// compute_attention (utils.py:229-235): attentions[b, :, query, 1:] -> (H, hf, wf) -> nearest x p.
// Token 1 + y*wf + x is patch (y, x): row-major flatten of the conv output (:131).
__global__ __launch_bounds__(256) void attn_map_kernel(const float *__restrict__ attn, float *__restrict__ maps,
                                                       int b, int H, int N, int query, int hf, int wf, int p) {
    const int W = wf * p, Hh = hf * p;
    const int64_t total = (int64_t)H * Hh * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % Hh), head = (int)(i / ((int64_t)W * Hh));
        const int tok = 1 + (y / p) * wf + (x / p);
        maps[i] = attn[(((int64_t)b * H + head) * N + query) * N + tok];
    }
}

hipError_t launch_attention_map(const float *attn, float *maps, int b, int heads, int n_tokens, int query, int hf,
                                int wf, int p, hipStream_t s) {
    const int64_t total = (int64_t)heads * hf * p * wf * p;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    attn_map_kernel<<<dim3(blocks), dim3(256), 0, s>>>(attn, maps, b, heads, n_tokens, query, hf, wf, p);
    return hipGetLastError();
}
