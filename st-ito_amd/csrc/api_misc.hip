// api_misc.hip -- error reporting and version for libstito_hip.so
#include "common.h"

namespace stito {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(256) void k_zero_words(unsigned *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

int zero_async(void *p, size_t bytes, hipStream_t st) {
    STITO_REQUIRE(((uintptr_t)p & 3) == 0 && bytes % 4 == 0, STITO_E_INVALID, "zero_async: %zu bytes at %p are not whole words", bytes, p);
    if (bytes == 0) return STITO_OK;
    const size_t n = bytes / 4;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, (unsigned *)p, n);
    STITO_LAUNCH_CHECK();
    return STITO_OK;
}

int device_info(DeviceInfo &info) {
    constexpr int MAX_DEV = 64;
    static DeviceInfo cache[MAX_DEV];
    static bool have[MAX_DEV];   // written once per device; a racing second writer stores the same values
    int dev = 0;
    STITO_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV || !have[dev]) {
        DeviceInfo d{};
        STITO_HIP_CHECK(hipDeviceGetAttribute(&d.cus, hipDeviceAttributeMultiprocessorCount, dev));
        STITO_HIP_CHECK(hipDeviceGetAttribute(&d.lds_per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        if (dev < 0 || dev >= MAX_DEV) { info = d; return STITO_OK; }
        cache[dev] = d;
        have[dev] = true;
    }
    info = cache[dev];
    return STITO_OK;
}
}  // namespace stito

extern "C" const char *stito_last_error(void) { return stito::g_err; }
extern "C" int stito_version(void) { return 10; }
