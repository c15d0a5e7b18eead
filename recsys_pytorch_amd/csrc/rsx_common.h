// rsx_common.h -- shared helpers of librsx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rsx.h"

#define RSX_API extern "C" __attribute__((visibility("default")))

void rsx_set_error(const char *fmt, ...);

#define RSX_CHECK_ARG(cond, msg)                                      \
    do {                                                              \
        if (!(cond)) {                                                \
            rsx_set_error("%s: invalid argument: %s", __func__, msg); \
            return RSX_E_INVALID;                                     \
        }                                                             \
    } while (0)

#define RSX_CHECK_LAUNCH()                                                           \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            rsx_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return RSX_E_HIP;                                                        \
        }                                                                            \
    } while (0)

static inline bool rsx_dim_ok(int d) { return d == 32 || d == 64 || d == 128; }

// number of CUs on the current device (cached)
int rsx_num_cus();

// fp32 hardware atomic add without return (global_atomic_add_f32); the file is
// built with -munsafe-fp-atomics so this never lowers to a CAS loop.
__device__ __forceinline__ void rsx_atomic_add(float *p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
