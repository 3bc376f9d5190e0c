// host_common.h — error reporting shared by the host-side translation units (engine.hip, swin_engine.hip):
// one thread-local message behind ocm_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/ocm_vit.h"

int ocm_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return ocm_fail(OCM_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------------
// kernel-class timing with hipEvents (ocm_prof_begin / ocm_prof_end, include/ocm_vit.h): PROF(class, stream) brackets the
// launches that follow it in its scope. Used by the ViT engine and the Swin engine (classes: the OCM_K_* of ocm_vit.h).
// ------------------------------------------------------------------------------------------
// One process-wide session (ocm_prof_begin .. ocm_prof_end). The state is shared by every handle and every host thread that
// launches through this library, so it is guarded: `on` is an atomic that launch paths read without the lock while no session
// is open (the common case: one relaxed load per launch); claiming a slot, recording into it and tearing the session down take
// the mutex, and a scope that outlives its session (ocm_prof_end on another thread between its two records) sees a new
// generation and records nothing.
struct Prof {
    std::atomic<bool> on{false};
    std::mutex mu;
    uint64_t gen = 0;
    uint32_t mask = 0;
    std::vector<hipEvent_t> ev;  // pairs
    std::vector<int> cls;
    size_t used = 0;             // pairs used
};
extern Prof g_prof;  // engine.hip

struct ProfScope {
    hipStream_t s;
    int idx = -1;
    uint64_t gen = 0;
    ProfScope(int kclass, hipStream_t stream) : s(stream) {
        if (!g_prof.on.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.on.load(std::memory_order_relaxed) && (g_prof.mask >> kclass & 1) && g_prof.used * 2 + 1 < g_prof.ev.size()) {
            idx = (int)g_prof.used++;
            gen = g_prof.gen;
            g_prof.cls[idx] = kclass;
            (void)hipEventRecord(g_prof.ev[2 * idx], s);
        }
    }
    ~ProfScope() {
        if (idx < 0) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.on.load(std::memory_order_relaxed) && g_prof.gen == gen) (void)hipEventRecord(g_prof.ev[2 * idx + 1], s);
    }
};
#define PROF(kclass, stream) ProfScope prof_scope_##__LINE__(kclass, stream)
