// Shared host-side helpers for libapx.so (error handling, launch checks).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/apx.h"

void apx_set_error(const char* fmt, ...);

#define APX_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            apx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return APX_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define APX_LAUNCH_CHECK()                                                                     \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess) {                                                               \
            apx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return APX_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

#define APX_REQUIRE(cond, msg)                                                                 \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            apx_set_error("invalid argument: %s (%s)", msg, #cond);                            \
            return APX_E_ARG;                                                                  \
        }                                                                                      \
    } while (0)

static inline int apx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// learner.hip, used by env.hip's apx_rollout (internal, not exported through include/apx.h)
int apx_mlp_forward_act(const float* params, int D, int H, int O, const float* x, int64_t B, const float* obs_mean, const float* obs_std, float* y, float* act,
                        const float* noise, float sigma, void* stream);
