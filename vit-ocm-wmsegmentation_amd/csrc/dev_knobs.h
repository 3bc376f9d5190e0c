// dev_knobs.h — development-only kernel-variant forcing (A/B microbenchmarks: tools/microbench_x3.py, tools/stamps_x3.py).
// The shipped library is built WITHOUT -DOCM_DEV: OCM_KNOB(i) is then the constant 0, every knob branch is dead code,
// the knob-only kernel instantiations are never emitted and `ocm_debug_knob` is not exported — dispatch depends on
// shape, precision, device and the per-handle options of include/ocm_vit.h only. `make dev` builds
// exp_libs/libocm_vit_dev.so with the knobs (declared in include/ocm_vit_dev.h; load it through OCM_VIT_LIB).
//   [0] nn.Linear GEMM variant, [1] write-through store mask (0 = shipped mask, -1 = none), [3] qkv GEMM variant,
//   [4] = 2 fused GEMM+LayerNorm on the LDS-DMA loop, [6] split-bf16 attention 1 = register-staged streaming kernel /
//   2 = whole-sequence kernel / 3 = the round-3 LDS-DMA loop on four waves (shipped: software-pipelined kernel up to 1 024 tokens,
//   LDS-DMA streaming kernel on eight waves above), [7] its 8-wave form 1 = never / 2 = always.
#pragma once
#ifdef OCM_DEV
extern int g_ocm_knobs[8];
#define OCM_KNOB(i) (g_ocm_knobs[(i)])
#else
#define OCM_KNOB(i) 0
#endif
