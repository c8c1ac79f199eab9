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

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace stito
