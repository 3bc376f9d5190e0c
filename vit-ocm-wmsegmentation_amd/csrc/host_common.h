// host_common.h — error reporting shared by the host-side translation units (engine.hip, swin_engine.hip):
// one thread-local message behind ocm_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ocm_vit.h"

int ocm_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return ocm_fail(OCM_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
