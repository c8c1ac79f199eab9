// Shared helpers for libstito_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/stito_hip.h"

namespace stito {

void set_error(const char *fmt, ...);

#define STITO_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::stito::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                               __FILE__, __LINE__);                                        \
            return STITO_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define STITO_LAUNCH_CHECK() STITO_HIP_CHECK(hipGetLastError())

#define STITO_REQUIRE(cond, code, ...)                                                     \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ::stito::set_error(__VA_ARGS__);                                               \
            return (code);                                                                 \
        }                                                                                  \
    } while (0)

// propagate a library status code (the callee has already set the error text)
#define STITO_TRY(expr)                        \
    do {                                       \
        const int _rc = (expr);                \
        if (_rc != STITO_OK) return _rc;       \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Zero `bytes` (a multiple of 4, 4-byte aligned) on `st` with a KERNEL, never hipMemsetAsync: a memset node captured into a hipGraph
// in front of a kernel that accumulates into the buffer (atomicMax of peaks / stream maxima) was not ordered in front of that kernel
// on replay (ROCm 7.2: every replay after the first kept the previous replay's maxima -- tools/graph_soak.py, 16 of 16 processes), and
// a kernel node is.  Same cost as the memset dispatch it replaces.
int zero_async(void *p, size_t bytes, hipStream_t st);

// Multiprocessor count and LDS per workgroup of the current device, queried once per device (not per launch).
struct DeviceInfo { int cus; int lds_per_block; };
int device_info(DeviceInfo &info);

}  // namespace stito
