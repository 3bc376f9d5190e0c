// kernels_gemm_inst.hip — explicit instantiations of gemm_kernels.h for ONE operand element type (OCM_INST_E: 0 = bf16,
// 1 = float, 2 = sp32) and ONE share of the entry points (OCM_INST_PART: 0 = nn.Linear, 1 = fused residual + LayerNorm, qkv
// projection, patch embedding, 2 = the activation-output epilogues of nn.Linear for sp32 only, split off part 0 because that
// object took twice as long as any other). The Makefile compiles this file seven times; nothing else instantiates the GEMM
// kernels.
#include "gemm_kernels.h"

#if OCM_INST_E == 0
typedef bf16 InstE;
#elif OCM_INST_E == 1
typedef float InstE;
#else
typedef sp32 InstE;
#endif

#if OCM_INST_PART == 2
template hipError_t launch_linear_mode<2, InstE>(const InstE *, const InstE *, const float *, const float *, void *, int, int, int,
                                                 hipStream_t, const LnFold &);
template hipError_t launch_linear_mode<3, InstE>(const InstE *, const InstE *, const float *, const float *, void *, int, int, int,
                                                 hipStream_t, const LnFold &);
#elif OCM_INST_PART == 0
#if OCM_INST_E == 2  // instantiated in part 2
extern template hipError_t launch_linear_mode<2, InstE>(const InstE *, const InstE *, const float *, const float *, void *, int,
                                                        int, int, hipStream_t, const LnFold &);
extern template hipError_t launch_linear_mode<3, InstE>(const InstE *, const InstE *, const float *, const float *, void *, int,
                                                        int, int, hipStream_t, const LnFold &);
#endif
#if defined(OCM_GEMM_STAMPS) && OCM_INST_E == 2
// development only: the cycle stamps of THIS translation unit's kernels (nn.Linear, split-bf16); g_stamps is per object
extern "C" int ocm_debug_stamps_linear(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * 8);
}
extern "C" int ocm_debug_wstamps_linear(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wstamps), (size_t)n * 8);
}
#endif
template hipError_t launch_linear_e<InstE>(const InstE *, const InstE *, const float *, const float *, void *, int, int, int,
                                           int, hipStream_t, const LnFold &, const StatsOut &);
#if OCM_INST_E != 2  // the strided launcher serves Swin (bf16 / fp32 only)
template hipError_t launch_linear_ld_e<InstE>(const InstE *, int64_t, const InstE *, const float *, const float *, void *,
                                              int64_t, int, int, int, int, hipStream_t);
#endif
#else
template hipError_t launch_resid_ln_e<InstE>(const InstE *, const InstE *, const float *, const float *, float *,
                                             const float *, const float *, void *, int, int, int, float, hipStream_t);
template hipError_t launch_qkv_e<InstE>(const InstE *, const InstE *, const float *, InstE *, InstE *, InstE *, float *, int,
                                        int, int, int, int, bool, hipStream_t, const LnFold &);
template hipError_t launch_patch_e<InstE>(const PatchArgs &, const InstE *, const float *, const float *, float *, int,
                                          hipStream_t, const StatsOut &);
#if defined(OCM_GEMM_STAMPS) && OCM_INST_E == 2
// development only: workgroups per CU the runtime grants a few of the shipped kernels (tools/stamps_x3.py)
extern "C" int ocm_debug_occupancy(int *out, int n) {
    int k = 0, v = 0;
    auto q = [&](const void *f, int threads, size_t lds) {
        v = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, f, threads, lds);
        if (k < n) out[k++] = v;
    };
    q((const void *)gemm_dma_kernel<Cfg64x128, sp32, false, 48, 2, EpiLinear<1, sp32>, 0>, 256, 2 * 192 * 128);
    q((const void *)gemm_dma_kernel<Cfg128x128, sp32, false, 12, 2, EpiLinear<2, sp32>, 0>, 256, 2 * 256 * 128);
    q((const void *)qkv_dma_kernel<Cfg128x128q, sp32, 12, 2>, 512, 2 * 256 * 128);
    q((const void *)gemm_kernel<GemmCfg<64, 384, 2, 4>, sp32, false, 48, RowLoader<sp32>, EpiResidLN<sp32, 384>>, 512,
      GemmCfg<64, 384, 2, 4>::LDS_BYTES);
    return k;
}
// development only: copy the cycle stamps of the last GEMM launches to the host (tools/stamps.py)
extern "C" int ocm_debug_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * 8);
}
#endif
#endif
